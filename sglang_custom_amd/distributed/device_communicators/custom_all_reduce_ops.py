"""The reference's ROCm custom-all-reduce op set on top of the mi355_ar_* C ABI.

The reference binds these names as torch ops (sgl-kernel/csrc/torch_extension_rocm.cc:25-55, Python
side sgl-kernel/python/sgl_kernel/allreduce.py:7-50) and drives them from CustomAllreduce
(python/sglang/srt/distributed/device_communicators/custom_all_reduce.py:282-300 set-up on HIP,
:384-393 buffer registration, :439-450 the two all-reduce forms, :500-506 tear-down).  The functions
here take the same arguments in the same order, so that class's HIP branch runs against this module
unchanged (`ops = custom_all_reduce_ops`).

Differences that follow from the one-launch design of the kernels underneath:
  * the "meta buffer" is the whole shared region (barrier flags + two staging areas); every call
    copies its input into the region inside the same launch, so no input buffer ever needs an IPC
    registration: register_buffer / register_graph_buffers accept what the reference passes and
    record it, get_graph_buffer_ipc_meta returns an empty list of addresses;
  * all_reduce_reg and all_reduce_unreg are the same launch (and the same bits).
"""
import ctypes
from typing import Dict, List, Tuple

import torch

from ... import _lib

_MAX_CAR_SIZE = 2 * 8192 * 1024   # custom_all_reduce.py:149-151: the largest message on ROCm

_DT = {torch.float32: _lib.F32, torch.bfloat16: _lib.BF16, torch.float16: _lib.F16}


class _Region:
    """Uncached device memory from mi355_ar_alloc_region, exported to torch through the CUDA array
    interface.  torch does not keep the exporting object alive, so the module does (_REGIONS) until
    free_meta_buffer / dispose."""

    def __init__(self, max_bytes: int):
        lib = _lib.load()
        ptr = ctypes.c_void_p()
        _lib.check(lib.mi355_ar_alloc_region(ctypes.byref(ptr), max_bytes), "mi355_ar_alloc_region")
        self.ptr, self.nbytes = ptr.value, int(lib.mi355_ar_region_bytes(max_bytes))
        self.__cuda_array_interface__ = {"shape": (self.nbytes,), "typestr": "|u1", "data": (self.ptr, False),
                                         "version": 2, "strides": None}

    def free(self):
        if self.ptr:
            _lib.load().mi355_ar_free_region(self.ptr)
            self.ptr = None


class _Comm:
    def __init__(self):
        self.handle = None
        self.opened: List[int] = []
        self.meta = None
        self.registered: List[Tuple[int, int]] = []     # (data_ptr, bytes) of buffers the caller registered
        self.graph_registrations = 0


_COMMS: Dict[int, _Comm] = {}
_REGIONS: Dict[int, _Region] = {}


def _signal_bytes() -> int:
    return int(_lib.load().mi355_ar_region_bytes(256)) - 512


def meta_size() -> int:
    """Bytes the region needs beyond `max_size`: the reference allocates meta_size() + max_size
    (custom_all_reduce.py:284).  The region holds the flags and TWO staging areas, so the second
    area (at the largest message the class ever asks for) is counted here."""
    return _signal_bytes() + _MAX_CAR_SIZE


def allocate_meta_buffer(size: int) -> torch.Tensor:
    """uint8 tensor over `size` (or slightly more) bytes of zeroed, uncached device memory on the
    current device (allocate_meta_buffer, custom_all_reduce.hip:156-172)."""
    payload = max(size - _signal_bytes(), 512)
    max_bytes = (payload // 2 + 255) // 256 * 256
    region = _Region(max_bytes)
    _REGIONS[region.ptr] = region
    t = torch.as_tensor(region, device=torch.device("cuda", torch.cuda.current_device()))
    assert t.data_ptr() == region.ptr
    return t


def free_meta_buffer(t: torch.Tensor) -> None:
    """free_meta_buffer (custom_all_reduce.hip:144; there the tensor's deleter): `t` must not be used
    afterwards."""
    region = _REGIONS.pop(t.data_ptr(), None)
    if region is not None:
        region.free()


def get_meta_buffer_ipc_handle(inp: torch.Tensor) -> torch.Tensor:
    """The 64-byte IPC handle of the allocation `inp` starts at, as a CPU uint8 tensor
    (get_meta_buffer_ipc_handle, custom_all_reduce.hip:146-154)."""
    lib = _lib.load()
    buf = (ctypes.c_uint8 * lib.mi355_ar_ipc_handle_bytes())()
    _lib.check(lib.mi355_ar_get_ipc_handle(inp.data_ptr(), buf), "mi355_ar_get_ipc_handle")
    return torch.tensor(list(buf), dtype=torch.uint8)


def init_custom_ar(meta: torch.Tensor, rank_data: torch.Tensor, handles: List[bytes], offsets: List[int],
                   rank: int, full_nvlink: bool) -> int:
    """init_custom_ar(meta, rank_data, handles, offsets, rank, full_nvlink) -> fa
    (custom_all_reduce.hip:13-51).  `rank_data` (the reference's table of registered pointers) is not
    needed; `full_nvlink` False is refused as there (:21-22 requires it for more than 2 ranks)."""
    lib = _lib.load()
    world = len(handles)
    if world > 8 or world % 2 == 1:
        raise ValueError("world size > 8 or odd is not supported")
    if world != len(offsets) or not 0 <= rank < world:
        raise ValueError("handles / offsets / rank mismatch")
    if world > 2 and not full_nvlink:
        raise ValueError("more than 2 ranks need a fully connected hive")
    max_bytes = (meta.numel() - _signal_bytes()) // 2 // 256 * 256
    c = _Comm()
    c.meta = meta
    hbytes = lib.mi355_ar_ipc_handle_bytes()
    ptrs = (ctypes.c_void_p * world)()
    for r in range(world):
        if r == rank:
            ptrs[r] = meta.data_ptr()
            continue
        p = ctypes.c_void_p()
        hb = (ctypes.c_uint8 * hbytes).from_buffer_copy(bytes(handles[r])[:hbytes])
        _lib.check(lib.mi355_ar_open_ipc_handle(hb, ctypes.byref(p)), "mi355_ar_open_ipc_handle")
        c.opened.append(p.value)
        ptrs[r] = p.value + int(offsets[r])
    comm = ctypes.c_void_p()
    _lib.check(lib.mi355_ar_init(ctypes.byref(comm), ptrs, rank, world, max_bytes), "mi355_ar_init")
    c.handle = comm.value
    _COMMS[c.handle] = c
    return c.handle


def _all_reduce(fa: int, inp: torch.Tensor, out: torch.Tensor) -> None:
    if inp.dtype not in _DT or out.dtype != inp.dtype or out.numel() != inp.numel():
        raise RuntimeError("custom allreduce currently supports float, half and bfloat16 of equal sizes")
    stream = torch.cuda.current_stream(inp.device).cuda_stream
    _lib.check(_lib.load().mi355_ar_all_reduce(fa, out.data_ptr(), inp.data_ptr(), inp.numel(), _DT[inp.dtype], 0,
                                               stream), "mi355_ar_all_reduce")


def all_reduce_reg(fa: int, inp: torch.Tensor, out: torch.Tensor) -> None:
    """all_reduce_reg(fa, inp, out!) (custom_all_reduce.hip:89-95)."""
    _all_reduce(fa, inp, out)


def all_reduce_unreg(fa: int, inp: torch.Tensor, reg_buffer: torch.Tensor, out: torch.Tensor) -> None:
    """all_reduce_unreg(fa, inp, reg_buffer, out!) (custom_all_reduce.hip:97-110): the reference first
    copies `inp` into the registered buffer; here the launch itself stages it."""
    if inp.numel() * inp.element_size() > reg_buffer.numel() * reg_buffer.element_size():
        raise RuntimeError("registered buffer is too small to contain the input")
    _all_reduce(fa, inp, out)


def register_buffer(fa: int, t: torch.Tensor, handles: List[bytes], offsets: List[int]) -> None:
    """register_buffer(fa, t, handles, offsets) (custom_all_reduce.hip:119-124): recorded only."""
    _COMMS[fa].registered.append((t.data_ptr(), t.numel() * t.element_size()))


def get_graph_buffer_ipc_meta(fa: int) -> Tuple[torch.Tensor, List[int]]:
    """(handle bytes, offsets) of the buffers a capture used (custom_all_reduce.hip:126-136): none."""
    return torch.empty(0, dtype=torch.uint8), []


def register_graph_buffers(fa: int, handles: List[bytes], offsets: List[List[int]]) -> None:
    """register_graph_buffers(fa, handles, offsets) (custom_all_reduce.hip:138-142): nothing to open."""
    _COMMS[fa].graph_registrations += 1


def dispose(fa: int) -> None:
    """dispose(fa) (custom_all_reduce.hip:112-115): peers' mappings closed, communicator freed."""
    c = _COMMS.pop(fa, None)
    if c is None:
        return
    lib = _lib.load()
    lib.mi355_ar_dispose(c.handle)
    for p in c.opened:
        lib.mi355_ar_close_ipc_handle(p)
    if c.meta is not None:
        free_meta_buffer(c.meta)
    c.meta = None
