"""The one-launch all-reduce (mi355_ar_*) with 2 and 4 ranks.

A test box has one GPU, so the ranks are separate processes that all use cuda:0: each allocates its
own region, the regions are opened through IPC exactly as between GPUs, and the kernels of the
ranks meet at the flag barriers while running side by side on the one device.  What this covers is
the protocol (barriers, index sets, both algorithms, in-place use, back-to-back calls, graph
replay) and the arithmetic, bit for bit against the oracle's rank-ordered fp32 sum; what it cannot
cover is xGMI itself.

The rank processes come from a fork server that is started while this module is imported, i.e.
before anything in the pytest process has touched the GPU: a process that has initialised the GPU
must not fork-and-exec on the test machines.
"""
import multiprocessing as mp
import os
import socket
import traceback

import pytest

_CTX = mp.get_context("forkserver")
_CTX.set_forkserver_preload([])
from multiprocessing import forkserver as _forkserver  # noqa: E402

_forkserver.ensure_running()

DTYPES = ("bfloat16", "float16", "float32")
# element counts: one 16-byte vector, fewer vectors than ranks, ragged slices, one decode step's
# hidden states at batch 1 / 64 (Llama-3-8B), and the largest the region takes
COUNTS = (8, 24, 1000 * 8, 4096, 64 * 4096 + 40, (1 << 20) - 8)
MAX_BYTES = 4 << 20


def _inputs(case, world, count, dtype):
    import torch

    g = torch.Generator().manual_seed(1000 * case + world)
    return [(torch.randn(count, generator=g) * (r + 1)).to(dtype) for r in range(world)]


def _rank_main(rank, world, port, queue, one_gpu_per_rank=False):
    try:
        import torch
        import torch.distributed as dist

        from oracle import ops_ref
        from sglang_custom_amd.distributed.device_communicators.custom_all_reduce import CustomAllreduce

        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        dev = rank if one_gpu_per_rank else 0
        torch.cuda.set_device(dev)
        ca = CustomAllreduce(dist.group.WORLD, dev, max_size=MAX_BYTES)
        assert not ca.disabled
        case = 0
        for dtn in DTYPES:
            dtype = getattr(torch, dtn)
            for count in COUNTS:
                if count * dtype.itemsize > MAX_BYTES or (count * dtype.itemsize) % 16:
                    continue
                for algo in (1, 2, 0):
                    case += 1
                    ins = _inputs(case, world, count, dtype)
                    x = ins[rank].cuda()
                    assert ca.should_custom_ar(x) == (x.numel() * x.element_size() < MAX_BYTES)
                    out = ca.all_reduce(x, algo=algo)
                    ca.check_status()
                    want = ops_ref.custom_all_reduce(ins)
                    assert torch.equal(out.cpu(), want), (dtn, count, algo)
                    assert torch.equal(x.cpu(), ins[rank]), "input modified"
        # back to back without host synchronisation, sizes and algorithms changing, in place
        pending = []
        for i, (count, algo) in enumerate([(4096, 1), (64 * 4096, 2), (8, 1), (64 * 4096, 1), (4096 * 3, 2)] * 4):
            ins = _inputs(500 + i, world, count, torch.bfloat16)
            x = ins[rank].cuda()
            ca.all_reduce(x, out=x, algo=algo)
            pending.append((x, ops_ref.custom_all_reduce(ins)))
        ca.check_status()
        for x, want in pending:
            assert torch.equal(x.cpu(), want)
        # all-reduce + residual add + RMSNorm in one launch == the two launches it replaces, bit for bit
        from sglang_custom_amd import ops

        for i, (dtn, rows, hidden) in enumerate((("bfloat16", 64, 4096), ("float16", 3, 512), ("float32", 5, 1024),
                                                 ("bfloat16", 70, 8192), ("bfloat16", 1, 4096))):
            dtype = getattr(torch, dtn)
            ins = _inputs(700 + i, world, rows * hidden, dtype)
            g = torch.Generator().manual_seed(7000 + i)
            res = torch.randn(rows, hidden, generator=g).to(dtype)
            wt = (1 + 0.1 * torch.randn(hidden, generator=g)).to(dtype).cuda()
            x, r1 = ins[rank].cuda().view(rows, hidden), res.cuda()
            assert ca.should_fuse_norm(x, hidden)
            ca.fused_add_rmsnorm(x, r1, wt, 1e-5)
            y, r2 = ca.all_reduce(ins[rank].cuda()).view(rows, hidden), res.cuda()
            ops.fused_add_rmsnorm(y, r2, wt, 1e-5)
            ca.check_status()
            assert torch.equal(x, y) and torch.equal(r1, r2), (dtn, rows, hidden)
            s64 = ops_ref.custom_all_reduce(ins).view(rows, hidden).double() + res.double()
            want = s64 * torch.rsqrt((s64 * s64).mean(-1, keepdim=True) + 1e-5) * wt.cpu().double()
            tol = 1e-5 if dtype == torch.float32 else 2 ** -7
            assert (x.cpu().double() - want).abs().max().item() <= tol * want.abs().max().item()
        # the reference's entry point: a new tensor, or None for what does not qualify
        ins = _inputs(900, world, 4096, torch.bfloat16)
        got = ca.custom_all_reduce(ins[rank].cuda())
        assert torch.equal(got.cpu(), ops_ref.custom_all_reduce(ins))
        assert ca.custom_all_reduce(torch.zeros(3, device="cuda", dtype=torch.bfloat16)) is None
        assert ca.custom_all_reduce(torch.zeros(MAX_BYTES, device="cuda", dtype=torch.bfloat16)) is None
        # inside a HIP graph: two launches per replay, replayed with new contents
        ins_a, ins_b = _inputs(901, world, 64 * 4096, torch.bfloat16), _inputs(902, world, 4096, torch.bfloat16)
        a, b = ins_a[rank].cuda(), ins_b[rank].cuda()
        oa, ob = torch.empty_like(a), torch.empty_like(b)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ca.all_reduce(a, out=oa)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        dist.barrier()
        graph = torch.cuda.CUDAGraph()
        with ca.capture(), torch.cuda.graph(graph):
            ca.all_reduce(a, out=oa)
            ca.all_reduce(b, out=ob)
        for rep in range(3):
            na, nb = _inputs(910 + rep, world, 64 * 4096, torch.bfloat16), _inputs(920 + rep, world, 4096, torch.bfloat16)
            a.copy_(na[rank])
            b.copy_(nb[rank])
            graph.replay()
            ca.check_status()
            assert torch.equal(oa.cpu(), ops_ref.custom_all_reduce(na))
            assert torch.equal(ob.cpu(), ops_ref.custom_all_reduce(nb))
        del graph
        ca.close()
        dist.destroy_process_group()
        queue.put((rank, "ok"))
    except BaseException:  # noqa: BLE001  (reported to the parent, which fails the test)
        queue.put((rank, traceback.format_exc()))


def _tp_rank_main(rank, world, port, queue, preset, dtn):
    """One tensor-parallel rank of the tiny Llama on cuda:0: HIP kernels on the rank's shard, the
    row-parallel sums and the logits gather through the one-launch collectives (also inside the
    decode graph), the process group itself on gloo."""
    try:
        import numpy as np
        import torch

        from oracle.llama_ref import OracleLlama
        from sglang_custom_amd import bench_one_batch as bob
        from sglang_custom_amd.configs.model_config import ModelConfig
        from sglang_custom_amd.distributed import parallel_state
        from sglang_custom_amd.managers.schedule_batch import Req
        from sglang_custom_amd.model_executor.model_runner import ModelRunner
        from sglang_custom_amd.model_loader.synthetic import hf_config_dict, make_synthetic_llama_weights
        from sglang_custom_amd.server_args import ServerArgs

        torch.cuda.set_device(0)
        dt = getattr(torch, dtn)
        cfg = hf_config_dict(preset, dtn)
        w = make_synthetic_llama_weights(cfg, seed=1234, dtype=dt)
        rng = np.random.RandomState(0)
        prompts = [rng.randint(0, cfg["vocab_size"], n).tolist() for n in (5, 17, 32)]
        sa = ServerArgs(dtype=dtn, max_total_tokens=512, max_running_requests=8, cuda_graph_bs=[4],
                        dist_backend="gloo", dist_init_method=f"tcp://127.0.0.1:{port}",
                        disable_custom_all_reduce=False)
        mr = ModelRunner(ModelConfig(cfg, dtype=dtn), sa, tp_rank=rank, tp_size=world,
                         weights={k: v.cuda() for k, v in w.items()})
        ca = parallel_state.get_custom_all_reduce()
        assert ca is not None and not ca.disabled
        # kept only because its start-up self-test agreed with the process group (gloo here) on every case
        assert parallel_state.custom_all_reduce_state() == "on (self-test passed)"
        # ... and a self-test against a reference that is NOT the sum names the first disagreement, on every rank
        why = ca.self_test(lambda t: t, lambda t: parallel_state.tensor_model_parallel_all_gather(t.unsqueeze(0), 0))
        assert why is not None and "all_reduce" in why, why
        assert mr.graph_runner is not None
        nxt, logits, batch = bob.extend([Req(i, p) for i, p in enumerate(prompts)], mr)
        toks, lgs = [nxt.cpu()], [logits.float().cpu()]
        for _ in range(3):
            nxt, logits = bob.decode(nxt, batch, mr)
            toks.append(nxt.cpu())
            lgs.append(logits.float().cpu())
        ca.check_status()
        # every row-parallel sum went through the fused all-reduce + norm launch (2 per layer per
        # forward, eager prefill + graph capture), the logits through the gather, none through gloo
        assert ca.calls["fused_add_rmsnorm"] >= 2 * cfg["num_hidden_layers"] * 2 and ca.calls["all_gather"] >= 2
        toks = torch.stack(toks, 1)
        if dt == torch.float32:
            ref_toks, ref_logits = OracleLlama(cfg, w, dt, max_total_tokens=512).generate_greedy(prompts, 4)
            assert torch.equal(toks, ref_toks), (toks, ref_toks)
            err = max((a - b.float()).abs().max().item() for a, b in zip(lgs, ref_logits))
            assert err < 1e-3, err
        else:
            # 16-bit: each rank rounds its partial sums before they are added, so the logits are
            # close to, not equal to, the one-process model's; the prefill step needs no teacher
            w32 = {k: v.float() for k, v in w.items()}
            cfg32 = hf_config_dict(preset, "float32")
            _, ref_logits = OracleLlama(cfg32, w32, torch.float32, max_total_tokens=512).generate_greedy(prompts, 1)
            err = (lgs[0] - ref_logits[0].float()).abs().max().item()
            assert err < 6e-2, err
        parallel_state.destroy_model_parallel()
        queue.put((rank, (toks.numpy(), torch.stack(lgs).numpy())))
    except BaseException:  # noqa: BLE001
        queue.put((rank, traceback.format_exc()))


@pytest.mark.gpu
@pytest.mark.parametrize("preset,dtn", [("tiny-llama-d64", "float32"), ("tiny-llama-d128", "bfloat16")])
def test_tensor_parallel_two_ranks_on_one_gpu(preset, dtn):
    import numpy as np

    port = _free_port()
    queue = _CTX.Queue()
    procs = [_CTX.Process(target=_tp_rank_main, args=(r, 2, port, queue, preset, dtn)) for r in range(2)]
    for p in procs:
        p.start()
    results = {}
    try:
        for _ in range(2):
            rank, msg = queue.get(timeout=140)
            results[rank] = msg
    finally:
        for p in procs:
            p.join(timeout=10)
            if p.is_alive():
                p.kill()
    for r in range(2):
        assert not isinstance(results[r], str), f"rank {r}: {results[r]}"
    # both ranks hold the same gathered logits and chose the same tokens
    assert np.array_equal(results[0][0], results[1][0])
    assert np.array_equal(results[0][1], results[1][1])


def _tp_rank_main_deepseek(rank, world, port, queue, preset, dtn):
    """One tensor-parallel rank of the tiny DeepSeek (MLA) stack on cuda:0: this rank's 8 of the 16 heads on the
    HIP kernels (multi-head prefill, absorbed decode over the replicated latent cache, eager and from a graph),
    o_proj / down_proj partial sums through the one-launch all-reduce."""
    try:
        import torch

        import cases
        from oracle.deepseek_ref import OracleDeepseek
        from sglang_custom_amd import bench_one_batch as bob
        from sglang_custom_amd.configs.model_config import ModelConfig
        from sglang_custom_amd.distributed import parallel_state
        from sglang_custom_amd.managers.schedule_batch import Req
        from sglang_custom_amd.mem_cache.memory_pool import MLATokenToKVPool
        from sglang_custom_amd.model_executor.model_runner import ModelRunner
        from sglang_custom_amd.model_loader.synthetic import hf_config_dict, make_synthetic_llama_weights
        from sglang_custom_amd.server_args import ServerArgs

        torch.cuda.set_device(0)
        dt = getattr(torch, dtn)
        cfg = hf_config_dict(preset, dtn)
        w = make_synthetic_llama_weights(cfg, seed=1234, dtype=dt)
        prompts = cases.e2e_prompts(cfg["vocab_size"], (5, 33, 70))
        sa = ServerArgs(dtype=dtn, max_total_tokens=1024, max_running_requests=8, cuda_graph_bs=[4],
                        dist_backend="gloo", dist_init_method=f"tcp://127.0.0.1:{port}",
                        disable_custom_all_reduce=False)
        mr = ModelRunner(ModelConfig(cfg, dtype=dtn), sa, tp_rank=rank, tp_size=world,
                         weights={k: v.cuda() for k, v in w.items()})
        assert isinstance(mr.token_to_kv_pool, MLATokenToKVPool) and mr.graph_runner is not None
        at = mr.model.model.layers[0].self_attn
        assert at.num_local_heads == cfg["num_attention_heads"] // world
        ca = parallel_state.get_custom_all_reduce()
        assert ca is not None and not ca.disabled
        steps = 4
        oracle = OracleDeepseek(cfg, w, dt, max_total_tokens=1024, max_reqs=16, context_len=512)
        ref_toks, ref_logits = oracle.generate_greedy(prompts, steps)
        nxt, logits, batch = bob.extend([Req(i, p) for i, p in enumerate(prompts)], mr)
        toks, lgs = [nxt.cpu()], [logits.float().cpu()]
        for s in range(steps - 1):
            feed = nxt if dt == torch.float32 else ref_toks[:, s].cuda()   # 16-bit: teacher-forced
            nxt, logits = bob.decode(feed, batch, mr)
            toks.append(nxt.cpu())
            lgs.append(logits.float().cpu())
        ca.check_status()
        assert ca.calls["all_reduce"] >= 2 * cfg["num_hidden_layers"]
        err = max((a - b.float()).abs().max().item() for a, b in zip(lgs, ref_logits))
        if dt == torch.float32:
            assert torch.equal(torch.stack(toks, 1), ref_toks)
            assert err < 1e-3, err
        else:
            assert err < 6e-2, err   # each rank rounds its partial sums before they are added
        parallel_state.destroy_model_parallel()
        queue.put((rank, (torch.stack(toks, 1).numpy(), torch.stack(lgs).numpy())))
    except BaseException:  # noqa: BLE001
        queue.put((rank, traceback.format_exc()))


@pytest.mark.gpu
@pytest.mark.parametrize("preset,dtn", [("tiny-deepseek", "float32"), ("tiny-deepseek-yarn", "bfloat16")])
def test_deepseek_tensor_parallel_two_ranks_on_one_gpu(preset, dtn):
    import numpy as np

    port = _free_port()
    queue = _CTX.Queue()
    procs = [_CTX.Process(target=_tp_rank_main_deepseek, args=(r, 2, port, queue, preset, dtn)) for r in range(2)]
    for p in procs:
        p.start()
    results = {}
    try:
        for _ in range(2):
            rank, msg = queue.get(timeout=140)
            results[rank] = msg
    finally:
        for p in procs:
            p.join(timeout=10)
            if p.is_alive():
                p.kill()
    for r in range(2):
        assert not isinstance(results[r], str), f"rank {r}: {results[r]}"
    assert np.array_equal(results[0][0], results[1][0])
    assert np.array_equal(results[0][1], results[1][1])


def _skew_rank_main(rank, world, port, queue):
    """Rank skew: a peer that is late but within the timeout changes nothing; a peer that stays away
    longer than the timeout makes the waiting rank's call fail loudly (NaN output + status error)."""
    try:
        import time

        import torch
        import torch.distributed as dist

        from oracle import ops_ref
        from sglang_custom_amd._lib import Mi355Error
        from sglang_custom_amd.distributed.device_communicators.custom_all_reduce import CustomAllreduce

        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        ca = CustomAllreduce(dist.group.WORLD, 0, max_size=MAX_BYTES)
        ins = _inputs(3000, world, 64 * 4096, torch.bfloat16)
        want = ops_ref.custom_all_reduce(ins)
        # (1) rank 1 arrives 2.5 s late: far inside the default timeout, results unchanged
        dist.barrier()
        if rank == 1:
            time.sleep(2.5)
        for algo in (1, 2):
            out = ca.all_reduce(ins[rank].cuda(), algo=algo)
            ca.check_status()
            assert torch.equal(out.cpu(), want), algo
        x, res = ins[rank].cuda().view(64, 4096), torch.zeros(64, 4096, dtype=torch.bfloat16, device="cuda")
        # (2) with a 300 ms timeout rank 1 stays away for 1.5 s: rank 0 gives up
        ca.set_timeout_ms(300)
        torch.cuda.synchronize()
        dist.barrier()
        if rank == 1:
            time.sleep(1.5)
        out = ca.all_reduce(ins[rank].cuda(), algo=1)
        torch.cuda.synchronize()
        if rank == 0:
            assert torch.isnan(out.float()).all(), "a sum over data that never arrived must be NaN"
            for wait in (True, False):
                try:
                    ca.check_status(wait)
                    raise AssertionError("check_status did not report the timed-out barrier")
                except Mi355Error as e:
                    assert "timed out" in str(e)
        else:
            # the late rank finds rank 0's flags and staged data in place: its own result is whole
            ca.check_status()
            assert torch.equal(out.cpu(), want)
        del x, res
        dist.barrier()
        ca.close()
        dist.destroy_process_group()
        queue.put((rank, "ok"))
    except BaseException:  # noqa: BLE001
        queue.put((rank, traceback.format_exc()))


@pytest.mark.gpu
def test_all_reduce_rank_skew_and_lost_peer():
    port = _free_port()
    queue = _CTX.Queue()
    procs = [_CTX.Process(target=_skew_rank_main, args=(r, 2, port, queue)) for r in range(2)]
    for p in procs:
        p.start()
    results = {}
    try:
        for _ in range(2):
            rank, msg = queue.get(timeout=100)
            results[rank] = msg
    finally:
        for p in procs:
            p.join(timeout=10)
            if p.is_alive():
                p.kill()
    assert results == {0: "ok", 1: "ok"}, "\n".join(f"rank {r}: {m}" for r, m in sorted(results.items()))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_all_reduce_ranks_on_one_gpu(world):
    port = _free_port()
    queue = _CTX.Queue()
    procs = [_CTX.Process(target=_rank_main, args=(r, world, port, queue)) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    try:
        for _ in range(world):
            rank, msg = queue.get(timeout=100)
            results[rank] = msg
    finally:
        for p in procs:
            p.join(timeout=10)
            if p.is_alive():
                p.kill()
    assert results == {r: "ok" for r in range(world)}, "\n".join(f"rank {r}: {m}" for r, m in sorted(results.items()))


def _ops_rank_main(rank, world, port, queue, one_gpu_per_rank=False):
    """The reference's CustomAllreduce HIP branch, call for call (custom_all_reduce.py:282-300, :384-393,
    :439-450, :500-506), against the op-set module; integer-valued inputs, so the sum is exact in every
    dtype and must equal the gloo sum bit for bit (as test/srt/test_custom_allreduce.py:58-170 compares
    with NCCL)."""
    try:
        import torch
        import torch.distributed as dist

        from sglang_custom_amd.distributed.device_communicators import custom_all_reduce_ops as ops

        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        dev = rank if one_gpu_per_rank else 0
        torch.cuda.set_device(dev)

        def gather(shard):
            data = [[None, None] for _ in range(world)]
            data[rank] = list(shard)
            for i in range(world):
                dist.broadcast_object_list(data[i], src=i, device="cpu")
            return [d[0] for d in data], [d[1] for d in data]

        max_size = 1 << 20
        meta = ops.allocate_meta_buffer(ops.meta_size() + max_size)
        assert meta.is_cuda and meta.dtype == torch.uint8 and meta.numel() >= ops.meta_size() + max_size
        assert int(meta.max()) == 0
        buffer = torch.empty(max_size, dtype=torch.uint8, device="cuda")
        handles, offsets = gather((bytes(ops.get_meta_buffer_ipc_handle(meta)), 0))
        rank_data = torch.empty(1024, dtype=torch.uint8, device="cuda")
        fa = ops.init_custom_ar(meta, rank_data, handles, offsets, rank, True)
        bh, bo = gather((bytes(ops.get_meta_buffer_ipc_handle(meta)), 0))
        ops.register_buffer(fa, buffer, bh, bo)
        dist.barrier()
        for dtype in (torch.float32, torch.float16, torch.bfloat16):
            for sz in (512, 2560, 32768, 262144, 524288):
                if sz * dtype.itemsize >= max_size:      # should_custom_ar (custom_all_reduce.py:414-437)
                    continue
                g = torch.Generator().manual_seed(sz + 7 * rank)
                inp = torch.randint(1, 16, (sz,), generator=g, dtype=torch.int32).to(dtype)
                want = inp.float().clone()
                dist.all_reduce(want)
                out = torch.empty(sz, dtype=dtype, device="cuda")
                ops.all_reduce_unreg(fa, inp.cuda(), buffer, out)          # eager form
                assert torch.equal(out.float().cpu(), want), ("unreg", dtype, sz)
                x = inp.cuda()
                g_out = torch.empty_like(x)
                torch.cuda.synchronize()
                dist.barrier()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):                              # graph form
                    ops.all_reduce_reg(fa, x, g_out)
                h, o = ops.get_graph_buffer_ipc_meta(fa)
                hs, os_ = gather((bytes(h), o))
                ops.register_graph_buffers(fa, hs, os_)
                dist.barrier()
                graph.replay()
                torch.cuda.synchronize()
                assert torch.equal(g_out.float().cpu(), want), ("reg", dtype, sz)
                del graph
                dist.barrier()
        try:
            ops.all_reduce_unreg(fa, torch.zeros(max_size, device="cuda"), buffer, torch.zeros(max_size, device="cuda"))
            raise AssertionError("an input larger than the registered buffer was accepted")
        except RuntimeError:
            pass
        dist.barrier()
        ops.dispose(fa)
        del meta
        dist.destroy_process_group()
        queue.put((rank, "ok"))
    except Exception:  # noqa: BLE001
        queue.put((rank, traceback.format_exc()))


def _run_ranks(target, world, *extra):
    port = _free_port()
    queue = _CTX.Queue()
    procs = [_CTX.Process(target=target, args=(r, world, port, queue) + extra) for r in range(world)]
    for p in procs:
        p.start()
    results = {}
    try:
        for _ in range(world):
            rank, msg = queue.get(timeout=150)
            results[rank] = msg
    finally:
        for p in procs:
            p.join(timeout=10)
            if p.is_alive():
                p.kill()
    assert results == {r: "ok" for r in range(world)}, "\n".join(f"rank {r}: {m}" for r, m in sorted(results.items()))


@pytest.mark.gpu
def test_reference_op_set_two_ranks_on_one_gpu():
    _run_ranks(_ops_rank_main, 2)


def _bench_launcher(queue, extra):
    """Runs inside a fork-server child (a process that has never touched the GPU): starts bench.py as the
    driver would, without a launcher, and hands its output back."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    try:
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(extra), env=env, capture_output=True,
                           text=True, timeout=420)
        queue.put((r.returncode, r.stdout, r.stderr[-4000:]))
    except Exception:  # noqa: BLE001
        queue.put((-1, "", traceback.format_exc()))


@pytest.mark.gpu
@pytest.mark.parametrize("sync_ids", [False, True])
def test_bench_tensor_parallel_branch_with_ranks_sharing_the_gpu(sync_ids):
    """bench.py's own TP branch end to end on the one GPU of the box: spawn two fresh ranks -> sharded model
    -> one-launch collectives (self-test against the gloo group) -> HIP-graph capture with the collectives
    inside -> timed loop -> ONE JSON line carrying the `collectives` object.  Llama-3.2-1B shapes, batch 8,
    128-token prompts.  With SYNC_TOKEN_IDS_ACROSS_TP the sampled ids also cross the MIN all-reduce."""
    import json

    queue = _CTX.Queue()
    args = ["--gpus", "2", "--ranks-share-gpu", "--model", "llama-3.2-1b", "--batch-size", "8", "--input-len", "128",
            "--steps", "6", "--warmup", "2", "--ttft-repeats", "1", "--no-cpu-baseline", "--custom-all-reduce", "auto"]
    if sync_ids:
        os.environ["SYNC_TOKEN_IDS_ACROSS_TP"] = "1"
    try:
        p = _CTX.Process(target=_bench_launcher, args=(queue, args))
        p.start()
        rc, out, err = queue.get(timeout=450)
        p.join(timeout=10)
    finally:
        os.environ.pop("SYNC_TOKEN_IDS_ACROSS_TP", None)
    assert rc == 0, err
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["scaling"] == "strong"
    assert line["value"] > 0 and abs(line["value"] - 8 * 6 / (line["ms_per_step"] * 6e-3)) < 1e-2 * line["value"]
    assert "REHEARSAL" in line["config"]["parallelism"] and line["config"]["hip_graph"] is True, line["config"]
    col = line["collectives"]
    assert col["ranks"] == 2 and col["all_reduces_per_step"] == 2 * 16 + 1
    assert col["custom_all_reduce"].startswith("on (self-test passed)"), col
    assert line["roofline"]["kernel"] == "gemm_ring_kernel" and 0 < line["roofline"]["frac"] < 1
    assert "cpu_baseline" not in line


def _hive_size():
    import torch

    n = torch.cuda.device_count()     # counting devices does not initialise the GPU in this process
    return 8 if n >= 8 else n // 2 * 2


@pytest.mark.gpu
@pytest.mark.skipif(_hive_size() < 2, reason="needs at least two GPUs of one xGMI hive")
def test_all_reduce_one_gpu_per_rank_over_xgmi():
    """One process per GPU, regions allocated uncached on each device and opened across devices: both
    algorithms and every size class bit for bit against the rank-ordered fp32 sum, then the
    reference's op set (eager and captured) against the host sum of integer-valued inputs."""
    world = _hive_size()
    _run_ranks(_rank_main, world, True)
    _run_ranks(_ops_rank_main, world, True)


def test_all_reduce_argument_checks_need_no_gpu():
    """Size rules of the entry points that return before touching a device."""
    from sglang_custom_amd import _lib

    lib = _lib.load()
    assert lib.mi355_ar_ipc_handle_bytes() == 64
    assert lib.mi355_ar_region_bytes(0) == -1 and lib.mi355_ar_region_bytes(100) == -1
    assert lib.mi355_ar_region_bytes(1 << 20) > 2 << 20
    # switch points: two ranks always one-shot; the reference's 512 KiB / 256 KiB for 4 / 8 ranks
    assert [lib.mi355_ar_choose_algo(2, b) for b in (16, 1 << 24)] == [1, 1]
    assert [lib.mi355_ar_choose_algo(4, b) for b in (512 * 1024 - 16, 512 * 1024)] == [1, 2]
    assert [lib.mi355_ar_choose_algo(8, b) for b in (256 * 1024 - 16, 256 * 1024)] == [1, 2]
    assert lib.mi355_ar_all_reduce(None, None, None, 8, _lib.BF16, 0, None) == -1
