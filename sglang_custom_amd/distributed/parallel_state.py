"""Tensor-parallel process group: one process per GPU, torch.distributed with the "nccl"
backend (= RCCL over xGMI on ROCm) for device tensors, "gloo" for CPU tests.

Covers what the hot path uses of the reference's GroupCoordinator
(python/sglang/srt/distributed/parallel_state.py:426-597) and communication_op.py:11-20:
sum all-reduce after row-parallel projections / the vocab-parallel embedding, and the
concatenating all-gather of vocab-sharded logits.  With world_size 1 every call is the identity.
"""
from typing import Optional

import torch
import torch.distributed as dist

_TP_GROUP: Optional[dist.ProcessGroup] = None
_TP_RANK = 0
_TP_SIZE = 1
_CA_COMM = None      # CustomAllreduce over the TP group, when enabled
_CPU_GROUP: Optional[dist.ProcessGroup] = None   # gloo group of the same ranks: host-side agreement, no device work
_CA_STATE = "off"
_BACKEND = "nccl"


def init_distributed_environment(world_size: int = 1, rank: int = 0, backend: str = "nccl",
                                 init_method: Optional[str] = None, use_custom_all_reduce: bool = False,
                                 local_device: int = 0, custom_all_reduce_self_test: bool = True) -> None:
    """Create the default process group (if world_size > 1) and use it as the TP group.

    use_custom_all_reduce: also build the one-launch xGMI all-reduce for tensors below its size
    limit (the reference's GroupCoordinator does the same, parallel_state.py:285-297); handles are
    exchanged over a gloo group created beside the device group.  All ranks must be in one node.
    custom_all_reduce_self_test: keep those kernels only if CustomAllreduce.self_test passes."""
    global _TP_GROUP, _TP_RANK, _TP_SIZE, _CA_COMM, _BACKEND, _CA_STATE, _CPU_GROUP
    _TP_RANK, _TP_SIZE, _BACKEND = rank, world_size, backend
    if world_size == 1:
        _TP_GROUP = None
        return
    if _CA_COMM is not None:       # a second set-up in one process: never leave the first communicator mapped
        _CA_COMM.close()
        _CA_COMM = None
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, init_method=init_method or "env://",
                                world_size=world_size, rank=rank)
    _TP_GROUP = dist.group.WORLD
    _CA_STATE = "off"
    if _CPU_GROUP is None:
        if backend == "gloo":
            _CPU_GROUP = _TP_GROUP
        else:
            try:
                _CPU_GROUP = dist.new_group(backend="gloo")
            except Exception as e:  # noqa: BLE001 -- e.g. no usable network interface for gloo on this machine
                import sys

                print(f"[parallel_state] no CPU side group ({type(e).__name__}: {e}); host-side agreement goes through "
                      "the device group, the one-launch collectives stay off", file=sys.stderr, flush=True)
                _CPU_GROUP = None
    if use_custom_all_reduce and _CPU_GROUP is None:
        _CA_STATE = "off (no CPU side group for the handle exchange)"
    elif use_custom_all_reduce:
        from sglang_custom_amd.distributed.device_communicators.custom_all_reduce import CustomAllreduce

        side = _CPU_GROUP
        try:
            _CA_COMM = CustomAllreduce(side, local_device)
        except Exception as e:  # noqa: BLE001 -- e.g. a peer's memory cannot be mapped on this machine
            _CA_COMM, _CA_STATE = None, f"off (set-up failed: {type(e).__name__}: {e})"
        if _CA_COMM is not None and _CA_COMM.disabled:
            _CA_COMM, _CA_STATE = None, f"off (world size {world_size} not supported)"
        if _CA_COMM is not None and custom_all_reduce_self_test:
            # used only after agreeing, on THIS machine and with THESE ranks, with the process group's
            # own collectives; otherwise every collective goes through the process group (RCCL)
            comm, _CA_COMM = _CA_COMM, None
            why = comm.self_test(tensor_model_parallel_all_reduce,
                                 lambda t: tensor_model_parallel_all_gather(t.unsqueeze(0), 0))
            if why is None:
                _CA_COMM, _CA_STATE = comm, "on (self-test passed)"
            else:
                import sys

                if rank == 0:
                    print(f"[parallel_state] custom all-reduce disabled: {why}", file=sys.stderr, flush=True)
                comm.close()
                _CA_STATE = f"off (self-test: {why})"
        elif _CA_COMM is not None:
            _CA_STATE = "on (no self-test)"


def custom_all_reduce_state() -> str:
    """"on (...)" / "off (...)": whether the one-launch collectives are in use, and why not."""
    return _CA_STATE


def get_custom_all_reduce():
    return _CA_COMM


def check_collectives(wait: bool = True) -> None:
    """Raise if any custom collective issued so far gave up waiting for a peer (its output was then
    filled with NaN).  Costs a stream synchronisation plus a 4-byte read-back, so the callers are the
    places that synchronise anyway: the per-step token read-back of the scheduler, the end of graph
    capture, the end of a benchmark's timed region.  wait=False skips the synchronisation (overlap
    scheduling: the step in question has been waited for, the next one is already running).  No-op
    without the custom collectives."""
    if _CA_COMM is not None and not _CA_COMM.disabled:
        _CA_COMM.check_status(wait)


def tp_barrier() -> None:
    """Host-side barrier over the TP ranks (no-op for one rank or a simulated shard).  Used before
    phases whose first device collective must not find the ranks seconds apart -- the warm-up runs
    of graph capture above all: the custom collectives give up after a bounded wait, and a timeout
    raises their status word for good (there is no reset; the communicator has to be dropped)."""
    if _TP_SIZE == 1 or _SIMULATED or _TP_GROUP is None:
        return
    if _CPU_GROUP is not None:
        dist.barrier(group=_CPU_GROUP)      # the CPU side group: no device work queued
    else:
        dist.barrier(group=_TP_GROUP, device_ids=[torch.cuda.current_device()])


def tp_any(flag: bool) -> bool:
    """True on every rank when `flag` is true on at least one (host-side, over the gloo group): how the
    ranks agree on a decision that one of them may have been forced into (a failed graph capture)."""
    if _TP_SIZE == 1 or _SIMULATED or _TP_GROUP is None:
        return bool(flag)
    if _CPU_GROUP is not None:
        t = torch.tensor([1 if flag else 0], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_CPU_GROUP)
    else:
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_TP_GROUP)
    return bool(t.item())


def drop_custom_all_reduce(reason: str) -> None:
    """Collective: every rank closes the one-launch collectives (after a barrier timeout their status
    word stays raised for good) and the process group's own collectives serve every later call."""
    global _CA_COMM, _CA_STATE
    if _CA_COMM is not None:
        _CA_COMM.close()
        _CA_COMM = None
    _CA_STATE = f"off ({reason})"


def sync_token_ids_across_tp(ids: torch.Tensor) -> torch.Tensor:
    """MIN all-reduce of the sampled ids over the TP group (layers/sampler.py:137-149 of the reference):
    ranks that disagree on a tie all continue with the smallest id instead of drifting apart."""
    if _TP_SIZE == 1 or _SIMULATED or _TP_GROUP is None:
        return ids
    if _BACKEND == "gloo" and ids.is_cuda:
        h = ids.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.MIN, group=_TP_GROUP)
        return ids.copy_(h)
    dist.all_reduce(ids, op=dist.ReduceOp.MIN, group=_TP_GROUP)
    return ids


_SIMULATED = False


def init_simulated_shard(world_size: int, rank: int = 0) -> None:
    """Measurement aid: build ONE rank's tensor-parallel shard on one GPU with no process group.
    The collectives become local stand-ins of the same output shape (all-reduce: identity,
    all-gather: the local shard repeated), so per-GPU kernel time of a TP configuration can be
    measured without the other ranks.  Results are NOT the model's outputs."""
    global _TP_GROUP, _TP_RANK, _TP_SIZE, _SIMULATED
    _TP_GROUP, _TP_RANK, _TP_SIZE, _SIMULATED = None, rank, world_size, True


def destroy_model_parallel() -> None:
    global _TP_GROUP, _TP_RANK, _TP_SIZE, _CA_COMM, _CPU_GROUP
    if _CA_COMM is not None:
        _CA_COMM.close()
    _TP_GROUP, _TP_RANK, _TP_SIZE, _CA_COMM, _CPU_GROUP = None, 0, 1, None, None


def get_tensor_model_parallel_world_size() -> int:
    return _TP_SIZE


def get_tensor_model_parallel_rank() -> int:
    return _TP_RANK


def get_tp_group() -> Optional[dist.ProcessGroup]:
    return _TP_GROUP


def tensor_model_parallel_all_reduce(x: torch.Tensor) -> torch.Tensor:
    """In-place sum over the TP group (communication_op.py:11-13)."""
    if _TP_SIZE == 1 or _SIMULATED:
        return x
    if _CA_COMM is not None and _CA_COMM.should_custom_ar(x):
        # decode-sized: one launch, fp32 sum in rank order (parallel_state.py:466-481 of the reference)
        return _CA_COMM.all_reduce(x, out=x)
    if _BACKEND == "gloo" and x.is_cuda:
        # rehearsal set-up (device tensors, CPU process group): stage through the host
        h = x.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=_TP_GROUP)
        return x.copy_(h)
    dist.all_reduce(x, op=dist.ReduceOp.SUM, group=_TP_GROUP)
    return x


def tensor_model_parallel_all_gather(x: torch.Tensor, dim: int = -1) -> torch.Tensor:
    """Concatenate the shards of every rank along `dim` (parallel_state.py:528-597)."""
    if _TP_SIZE == 1:
        return x
    if dim < 0:
        dim += x.dim()
    if _SIMULATED:
        return torch.cat([x] * _TP_SIZE, dim=dim)
    x = x.contiguous()
    if _CA_COMM is not None and x.is_cuda and _CA_COMM.should_custom_ag(x):
        out = _CA_COMM.all_gather(x).movedim(0, dim)
        shape = list(x.shape)
        shape[dim] *= _TP_SIZE
        return out.reshape(shape)
    if _BACKEND == "gloo" and x.is_cuda:
        h = x.cpu()
        out = torch.empty((_TP_SIZE * h.shape[0],) + tuple(h.shape[1:]), dtype=h.dtype)
        dist.all_gather_into_tensor(out, h, group=_TP_GROUP)
        out = out.to(x.device).view((_TP_SIZE,) + tuple(x.shape)).movedim(0, dim)
        shape = list(x.shape)
        shape[dim] *= _TP_SIZE
        return out.reshape(shape)
    # rank-major concatenation along dim 0: the one output layout every backend accepts
    out = torch.empty((_TP_SIZE * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x, group=_TP_GROUP)
    out = out.view((_TP_SIZE,) + tuple(x.shape)).movedim(0, dim)
    shape = list(x.shape)
    shape[dim] *= _TP_SIZE
    return out.reshape(shape)
