"""One-launch sum all-reduce over the GPUs of an xGMI hive for decode-sized tensors.

Mirrors the reference's CustomAllreduce
(python/sglang/srt/distributed/device_communicators/custom_all_reduce.py:146-509): same
constructor (a non-NCCL process group for the handle exchange, a device, max_size), the same
`should_custom_ar` rule (byte size a multiple of 16, contiguous, below max_size), and
`custom_all_reduce(x)` returning a new tensor or None when the tensor does not qualify.

What differs is underneath: the kernels copy their input into the rank's own IPC region as the
first stage of the launch, so there is no buffer registration, no separate device copy outside
graphs and nothing to collect after a graph capture (`capture()` is kept as a no-op context for
call-site compatibility with parallel_state.py:377-378 of the reference).
"""
import ctypes
from contextlib import contextmanager
from typing import List, Optional

import torch
import torch.distributed as dist

from sglang_custom_amd import _lib

_DT = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}


def is_weak_contiguous(inp: torch.Tensor) -> bool:
    """Contiguous, or a view covering its whole storage tail (custom_all_reduce.py:139-143)."""
    return inp.is_contiguous() or (
        inp.untyped_storage().nbytes() - inp.storage_offset() * inp.element_size()
        == inp.numel() * inp.element_size())


class CustomAllreduce:
    _SUPPORTED_WORLD_SIZES = [2, 4, 6, 8]
    # the reference's ROCm crossover to the ring collective (custom_all_reduce.py:149-151)
    _MAX_CAR_SIZE = 2 * 8192 * 1024

    def __init__(self, group: dist.ProcessGroup, device, max_size: int = _MAX_CAR_SIZE) -> None:
        self.disabled = True
        self._comm = None
        self._region = None
        self._opened: List[int] = []
        self._IS_CAPTURING = False
        self.calls = {"all_reduce": 0, "all_gather": 0, "fused_add_rmsnorm": 0}   # launches issued (incl. captured)
        self.group = group
        assert dist.get_backend(group) != dist.Backend.NCCL, \
            "CustomAllreduce exchanges IPC handles as Python objects: attach it to a non-NCCL group"
        rank = dist.get_rank(group=group)
        world_size = dist.get_world_size(group=group)
        if world_size == 1 or world_size not in self._SUPPORTED_WORLD_SIZES:
            return
        if isinstance(device, int):
            device = torch.device(f"cuda:{device}")
        elif isinstance(device, str):
            device = torch.device(device)
        self.device = device
        self.rank, self.world_size, self.max_size = rank, world_size, int(max_size)
        lib = _lib.load()
        self._lib = lib
        # Set-up is collective: a rank whose step fails must not leave the others waiting in the next
        # exchange, so every step's outcome is gathered and all ranks raise together.
        err, handle = None, b""
        hbytes = lib.mi355_ar_ipc_handle_bytes()
        try:
            with torch.cuda.device(device):
                region = ctypes.c_void_p()
                _lib.check(lib.mi355_ar_alloc_region(ctypes.byref(region), self.max_size), "mi355_ar_alloc_region")
                self._region = region.value
                buf = (ctypes.c_uint8 * hbytes)()
                _lib.check(lib.mi355_ar_get_ipc_handle(self._region, buf), "mi355_ar_get_ipc_handle")
                handle = bytes(buf)
        except Exception as e:  # noqa: BLE001
            err = f"rank {rank}: {type(e).__name__}: {e}"
        gathered = self._gather((handle, err))
        self._raise_together([g[1] for g in gathered])
        try:
            with torch.cuda.device(device):
                ptrs = (ctypes.c_void_p * world_size)()
                for r, (h, _) in enumerate(gathered):
                    if r == rank:
                        ptrs[r] = self._region
                    else:
                        p = ctypes.c_void_p()
                        hb = (ctypes.c_uint8 * hbytes).from_buffer_copy(h)
                        _lib.check(lib.mi355_ar_open_ipc_handle(hb, ctypes.byref(p)), "mi355_ar_open_ipc_handle")
                        ptrs[r] = p.value
                        self._opened.append(p.value)
                comm = ctypes.c_void_p()
                _lib.check(lib.mi355_ar_init(ctypes.byref(comm), ptrs, rank, world_size, self.max_size), "mi355_ar_init")
                self._comm = comm.value
        except Exception as e:  # noqa: BLE001
            err = f"rank {rank}: {type(e).__name__}: {e}"
        self._raise_together(self._gather(err))
        # every rank's region is zeroed and mapped everywhere before the first launch
        dist.barrier(group=group)
        self.disabled = False

    def _raise_together(self, errors) -> None:
        first = next((e for e in errors if e), None)
        if first is None:
            return
        for p in self._opened:
            self._lib.mi355_ar_close_ipc_handle(p)
        if self._comm:
            self._lib.mi355_ar_dispose(self._comm)
        if self._region:
            self._lib.mi355_ar_free_region(self._region)
        self._comm, self._region, self._opened = None, None, []
        raise RuntimeError(f"custom all-reduce set-up failed on at least one rank ({first})")

    def _gather(self, obj):
        """all_gather of a Python object by per-rank broadcasts (all_gather_object does not work on
        gloo under inference mode; the reference does the same, custom_all_reduce.py:360-382)."""
        out = []
        for src in sorted(dist.get_process_group_ranks(self.group)):
            box = [obj if dist.get_rank() == src else None]
            dist.broadcast_object_list(box, src=src, group=self.group, device="cpu")
            out.append(box[0])
        return out

    @contextmanager
    def capture(self):
        """Graph capture needs no registration here; kept for the reference's call sites."""
        self._IS_CAPTURING = True
        try:
            yield
        finally:
            self._IS_CAPTURING = False

    def should_custom_ar(self, inp: torch.Tensor) -> bool:
        if self.disabled:
            return False
        inp_size = inp.numel() * inp.element_size()
        if inp_size == 0 or inp_size % 16 != 0 or inp.dtype not in _DT:
            return False
        if not is_weak_contiguous(inp):
            return False
        return inp_size < self.max_size

    def all_reduce(self, inp: torch.Tensor, *, out: Optional[torch.Tensor] = None, algo: int = 0) -> torch.Tensor:
        """out = sum over ranks of inp (out may be inp). algo 0 chooses by size, 1 one-shot, 2 two-shot."""
        if out is None:
            out = torch.empty_like(inp)
        stream = torch.cuda.current_stream(inp.device).cuda_stream
        self.calls["all_reduce"] += 1
        _lib.check(self._lib.mi355_ar_all_reduce(self._comm, out.data_ptr(), inp.data_ptr(), inp.numel(),
                                                 _DT[inp.dtype], algo, stream), "mi355_ar_all_reduce")
        return out

    def custom_all_reduce(self, input: torch.Tensor) -> Optional[torch.Tensor]:
        """The reference's entry point (custom_all_reduce.py:475-498): None when the tensor does not
        qualify, else a new tensor holding the sum."""
        if self.disabled or not self.should_custom_ar(input):
            return None
        return self.all_reduce(input)

    def should_custom_ag(self, inp: torch.Tensor) -> bool:
        size = inp.numel() * inp.element_size()
        return (not self.disabled) and size > 0 and size % 16 == 0 and size <= self.max_size and inp.is_contiguous()

    def all_gather(self, inp: torch.Tensor) -> torch.Tensor:
        """[world, *inp.shape]: every rank's tensor, rank-major (one launch, any dtype)."""
        out = torch.empty((self.world_size,) + tuple(inp.shape), dtype=inp.dtype, device=inp.device)
        stream = torch.cuda.current_stream(inp.device).cuda_stream
        self.calls["all_gather"] += 1
        _lib.check(self._lib.mi355_ar_all_gather(self._comm, out.data_ptr(), inp.data_ptr(),
                                                 inp.numel() * inp.element_size(), stream), "mi355_ar_all_gather")
        return out

    def should_fuse_norm(self, x: torch.Tensor, hidden: int) -> bool:
        ve = 16 // x.element_size()
        return (self.should_custom_ar(x) and x.dim() == 2 and x.is_contiguous() and x.shape[1] == hidden
                and hidden % ve == 0 and hidden // ve <= 2048)

    def fused_add_rmsnorm(self, x: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float) -> None:
        """In place: x (this rank's partial sums) -> RMSNorm(sum over ranks + residual) * weight,
        residual -> sum + residual; same bits as all_reduce followed by the fused add+RMSNorm."""
        assert residual.is_contiguous() and residual.shape == x.shape and residual.dtype == x.dtype
        stream = torch.cuda.current_stream(x.device).cuda_stream
        self.calls["fused_add_rmsnorm"] += 1
        _lib.check(self._lib.mi355_ar_fused_add_rmsnorm(self._comm, x.data_ptr(), residual.data_ptr(), weight.data_ptr(),
                                                        x.shape[0], x.shape[1], float(eps), _DT[x.dtype], stream),
                   "mi355_ar_fused_add_rmsnorm")

    def self_test(self, reference_all_reduce, reference_all_gather) -> Optional[str]:
        """Run every collective of this class once per algorithm and size class on integer-valued
        inputs (sums exact in bf16, so any correct all-reduce gives the same bits) and compare with the
        process group's own collectives, passed in as callables.  Returns None when every rank agreed
        on every case, else the first reason (the same string on every rank).  A peer that never
        arrives costs one 2-second barrier timeout, not a hang."""
        reason = None
        try:
            self.set_timeout_ms(2000)
            g = torch.Generator().manual_seed(1234 + self.rank)
            for count, algo in ((4096, 1), (4096, 2), (64 * 4096, 1), (64 * 4096, 2), (1 << 20, 0)):
                if count * 2 >= self.max_size:
                    continue
                x = torch.randint(-8, 9, (count,), generator=g, dtype=torch.int32).to(torch.bfloat16).to(self.device)
                want = reference_all_reduce(x.clone())
                got = self.all_reduce(x, algo=algo)
                if not torch.equal(got, want):
                    reason = reason or f"all_reduce of {count} bf16 (algo {algo}) differs from the process group's sum"
            x = torch.randint(-8, 9, (64, 2048), generator=g, dtype=torch.int32).to(torch.bfloat16).to(self.device)
            if self.should_custom_ag(x) and not torch.equal(self.all_gather(x), reference_all_gather(x)):
                reason = reason or "all_gather differs from the process group's"
            res = torch.randint(-8, 9, (64, 2048), generator=g, dtype=torch.int32).to(torch.bfloat16).to(self.device)
            w = torch.ones(2048, dtype=torch.bfloat16, device=self.device)
            if self.should_fuse_norm(x, 2048):
                from ...ops_backend import get_ops

                a, ra = self.all_reduce(x), res.clone()
                get_ops().fused_add_rmsnorm(a, ra, w, 1e-5)
                b, rb = x.clone(), res.clone()
                self.fused_add_rmsnorm(b, rb, w, 1e-5)
                if not (torch.equal(a, b) and torch.equal(ra, rb)):
                    reason = reason or "all-reduce fused with add+RMSNorm differs from the two steps"
            self.check_status()
        except Exception as e:  # noqa: BLE001 -- any failure means: do not use these kernels here
            reason = reason or f"{type(e).__name__}: {e}"
        finally:
            try:
                self.set_timeout_ms(20000)
            except Exception:  # noqa: BLE001
                pass
        reasons = self._gather(reason)
        return next((r for r in reasons if r), None)

    def set_timeout_ms(self, ms: int) -> None:
        """How long a barrier may wait for a peer before the call gives up (fills its output with NaN
        and makes check_status raise); applies to launches and graph captures made afterwards."""
        _lib.check(self._lib.mi355_ar_set_timeout_ms(self._comm, int(ms)), "mi355_ar_set_timeout_ms")

    def check_status(self, wait: bool = True) -> None:
        """Raises if any barrier wait has timed out.  A raised status is terminal for this communicator:
        the counter is never reset (the ranks' barrier epochs no longer agree after a timeout), so the
        owner must close() it and build a new one, or fall back to the process group's collectives.
        wait=True synchronises the current stream first;
        wait=False reads the counter beside whatever is running (the caller has already waited for the
        work it asks about)."""
        if wait:
            stream = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(self._lib.mi355_ar_status(self._comm, stream), "mi355_ar_status")
        else:
            _lib.check(self._lib.mi355_ar_poll_status(self._comm), "mi355_ar_poll_status")

    def close(self) -> None:
        if self._comm:
            torch.cuda.synchronize(self.device)
            # nobody unmaps a region that a peer's last launch may still be reading
            dist.barrier(group=self.group)
            self._lib.mi355_ar_dispose(self._comm)
            for p in self._opened:
                self._lib.mi355_ar_close_ipc_handle(p)
            self._lib.mi355_ar_free_region(self._region)
            self._comm, self._region, self._opened = None, None, []
        self.disabled = True
