"""Headline benchmark: decode output tokens/s (+ TTFT) of Llama-3-8B bf16, static batch of 64
requests, 1024-token prompts, on MI355X -- BASELINE.json configs[1] -- measured with the
semantics of the reference's bench_one_batch.py (synthetic fixed-length prompts, greedy).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

One "step" = one decode forward of the whole batch (bs new tokens).  W untimed warm-up steps,
then exactly K timed steps between barrier + synchronize; rank 0 prints ONE JSON line.
N > 1: one process per GPU.  Started under torchrun the ranks are torchrun's; started as a plain
`python bench.py --gpus N`, N fresh child processes are spawned BEFORE anything touches the GPU
(as the reference starts one process per tensor-parallel rank, bench_one_batch.py:524-544), and the
run fails if fewer than N ranks come up.  Default `--parallelism tp`: ONE model sharded over the N
GPUs (heads / MLP columns / vocabulary, two sum-all-reduces per layer over RCCL, or the one-launch
xGMI kernels when their start-up self-test passes, see --custom-all-reduce), the global batch fixed ("strong" scaling); `--parallelism dp`
runs N independent TP=1 replicas instead ("weak").  `n_gpus` in the line is the number of ranks that
actually ran.

The same line carries
  roofline      -- the dominant kernel (weight-streaming skinny GEMM) on rank 0's shard: algorithmic
                   bytes per launch / its average duration measured here with HIP events on the
                   launch stream, against the 8 TB/s HBM3E peak;
  cpu_baseline  -- the CPU restatement (oracle/) of the same decode step timed on this host's cores
                   on a bounded sample (a few of the layers + the LM head, scaled), N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E datasheet peak


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--model", default="llama-3-8b")
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--batch-size", type=int, default=64)
    ap.add_argument("--input-len", type=int, default=1024)
    ap.add_argument("--quantization", default=None, choices=[None, "w8a8_fp8"])
    ap.add_argument("--kv-cache-dtype", default="auto", choices=["auto", "fp8_e4m3"],
                    help="fp8_e4m3: optional reduced-precision KV cache (NOT the headline configuration)")
    ap.add_argument("--parallelism", default="tp", choices=["dp", "tp"])
    ap.add_argument("--custom-all-reduce", choices=["auto", "on", "off"], default="auto",
                    help="with --parallelism tp: row-parallel sums and the logits gather through the one-launch "
                         "xGMI collectives.  auto (default): only if their start-up self-test agrees bit for bit "
                         "with RCCL on this machine; on: without the self-test; off: RCCL only")
    ap.add_argument("--simulate-tp", type=int, default=0,
                    help="measurement aid (not the contract metric): run ONE rank's shard of a TP=N "
                         "configuration on one GPU with the collectives stubbed out")
    ap.add_argument("--kv-layout", default="auto", choices=["auto", "token_major", "head_major"],
                    help="memory layout of the multi-head KV pool (auto: head-major on the GPU for 16-bit models)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--ttft-repeats", type=int, default=3)
    ap.add_argument("--prefill-gemm", default=None, choices=["auto", "library", "tiled"],
                    help="prefill (M > 64) 16-bit GEMMs: tiled (default: the hand-written 256x256-tile kernel everywhere), "
                         "auto (hand-written gate_up with fused SiLU-mul, hipBLASLt for the plain projections, which it "
                         "runs 4-6 %% faster), or hipBLASLt everywhere")
    ap.add_argument("--ranks-share-gpu", action="store_true",
                    help="rehearsal on a one-GPU box: every rank runs on GPU 0 (process group on gloo, since RCCL "
                         "refuses two ranks per device; the one-launch collectives 'on' with their self-test "
                         "against that group).  Exercises spawn -> TP model -> graph capture -> timed loop -> "
                         "the collectives object; its tok/s is NOT a multi-GPU number and the line says so")
    ap.add_argument("--dry-run", action="store_true",
                    help="rendezvous rehearsal without a GPU: every rank joins a gloo group, meets at the "
                         "barriers and rank 0 prints the line's bookkeeping fields (value null)")
    return ap.parse_args()


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N fresh interpreters, one per GPU, with the
    rendezvous variables torchrun would set.  This process never touches the GPU; it relays the
    children's output and returns non-zero if any rank failed (the others are then stopped)."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = set(range(n))
    while alive:
        for r in list(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code
                print(f"[bench] rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                for o in alive:
                    procs[o].terminate()
        time.sleep(0.05)
    return rc


def weight_bytes_per_step(cfg, tp, dtype_bytes, quant):
    """Bytes of every matmul weight touched once per decode step on one GPU (layers + LM head)."""
    if "kv_lora_rank" in cfg:   # multi-head latent attention (dense DeepSeek layers)
        h, H = cfg["hidden_size"], cfg["num_attention_heads"] // tp
        qr, kr, rope = cfg.get("q_lora_rank") or 0, cfg["kv_lora_rank"], cfg["qk_rope_head_dim"]
        nope, vd = cfg["qk_nope_head_dim"], cfg["v_head_dim"]
        inter, vocab, L = cfg["intermediate_size"] // tp, cfg["vocab_size"] // tp, cfg["num_hidden_layers"]
        a_proj = (qr + kr + rope) * h + H * (nope + rope) * qr if qr else H * (nope + rope) * h + (kr + rope) * h
        per_layer = a_proj + h * H * vd + 3 * inter * h
        absorbed = H * (nope + vd) * kr   # w_kc / w_vc: kv_b_proj folded into q and out of the attention output, 16-bit
        wb = 1 if quant == "w8a8_fp8" else dtype_bytes
        return L * (per_layer * wb + absorbed * dtype_bytes) + vocab * h * dtype_bytes, L * (per_layer + absorbed) + vocab * h
    h, d = cfg["hidden_size"], cfg["head_dim"]
    hq, hkv = cfg["num_attention_heads"] // tp, max(1, cfg["num_key_value_heads"] // tp)
    inter, vocab, L = cfg["intermediate_size"] // tp, cfg["vocab_size"] // tp, cfg["num_hidden_layers"]
    per_layer = (hq + 2 * hkv) * d * h + hq * d * h + 2 * inter * h + inter * h
    wb = 1 if quant == "w8a8_fp8" else dtype_bytes
    return L * per_layer * wb + vocab * h * dtype_bytes, L * per_layer + vocab * h


def gemm_shapes(cfg, tp):
    h, d = cfg["hidden_size"], cfg["head_dim"]
    hq, hkv = cfg["num_attention_heads"] // tp, max(1, cfg["num_key_value_heads"] // tp)
    inter = cfg["intermediate_size"] // tp
    return [((hq + 2 * hkv) * d, h), (h, hq * d), (2 * inter, h), (h, inter)]


def measure_dominant_kernel(mr, bs, cfg, tp, dtype_bytes, quant=None):
    """Average duration of the decode projection kernel (gemm_ring_kernel; its fp8 instances under
    w8a8_fp8) over one step's worth of launches -- every layer's real weights in order, M = bs, through
    the same call the model makes -- bracketed by HIP events on the stream the kernels are launched on."""
    from sglang_custom_amd.layers.quantization.unquant import (QuantizedAct, decode_linear, decode_linear_deferred,
                                                              gate_up_silu_mul)
    from sglang_custom_amd.ops_backend import get_ops

    dev = mr.device
    layers = mr.model.model.layers
    x_h = torch.randn(bs, cfg["hidden_size"], device=dev).to(mr.dtype)
    x_attn = torch.randn(bs, layers[0].self_attn.o_proj.weight.shape[1], device=dev).to(mr.dtype)
    x_mlp = torch.randn(bs, layers[0].mlp.down_proj.weight.shape[1], device=dev).to(mr.dtype)
    head = mr.model.lm_head
    if quant == "w8a8_fp8":
        # the decode path hands the fp8 projections rows that their producers already quantised
        ops = get_ops()
        q_h, q_attn, q_mlp = (QuantizedAct(*ops.per_token_quant_fp8(x), mr.dtype) for x in (x_h, x_attn, x_mlp))

    def proj(layer, x, needs_all_reduce=True):
        # the call the decode path makes: split-K partials left to the consuming kernel when the
        # layer qualifies, the in-kernel combine otherwise
        if decode_linear_deferred(layer, x, needs_all_reduce=needs_all_reduce) is None:
            if isinstance(x, QuantizedAct):
                layer.quant_method.apply(layer, x, None)
            else:
                decode_linear(layer, x, None)

    def one_pass():
        for layer in layers:
            if quant == "w8a8_fp8":
                proj(layer.self_attn.qkv_proj, q_h, needs_all_reduce=False)
                proj(layer.self_attn.o_proj, q_attn)
                gu = layer.mlp.gate_up_proj
                gu.quant_method.apply(gu, q_h, None)
                proj(layer.mlp.down_proj, q_mlp)
            else:
                proj(layer.self_attn.qkv_proj, x_h, needs_all_reduce=False)
                proj(layer.self_attn.o_proj, x_attn)
                gate_up_silu_mul(layer.mlp.gate_up_proj, layer.mlp.act_fn, x_h)
                proj(layer.mlp.down_proj, x_mlp)
        decode_linear(head, x_h, None, out_dtype=torch.float32)

    stream = torch.cuda.current_stream()
    one_pass()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()   # replay from a graph: no host launch gaps inside the bracket
    with torch.cuda.graph(g):
        one_pass()
    reps = 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g.replay()
    torch.cuda.synchronize()
    e0.record(stream)
    for _ in range(reps):
        g.replay()
    e1.record(stream)
    torch.cuda.synchronize()
    launches = 4 * len(layers) + 1
    total_ms = e0.elapsed_time(e1) / reps
    wbytes, _ = weight_bytes_per_step(cfg, tp, dtype_bytes, quant)
    # algorithmic bytes: weights once (fp8: one byte each, plus the per-channel scales) + activations in
    # (fp8: one byte each plus per-token scales) / out (16-bit, or fp32 partial sums are NOT counted: they are
    # this implementation's choice, not the algorithm's)
    in_b = 1 if quant == "w8a8_fp8" else dtype_bytes
    act = sum(bs * (k * in_b + n * dtype_bytes) for (n, k) in gemm_shapes(cfg, tp)) * len(layers) \
        + bs * (cfg["hidden_size"] * dtype_bytes + cfg["vocab_size"] // tp * 4)
    if quant == "w8a8_fp8":
        act += sum(4 * (n + bs) for (n, _) in gemm_shapes(cfg, tp)) * len(layers)
    bytes_per_launch = (wbytes + act) / launches
    dur_s = total_ms * 1e-3 / launches
    return dict(kernel="gemm_ring_kernel", launches_per_step=launches,
                avg_duration_us=dur_s * 1e6, bytes_per_launch=bytes_per_launch,
                achieved_gbs=bytes_per_launch / dur_s / 1e9)


def committed_pmc_traffic(kernel: str, quant=None):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 --pmc FETCH_SIZE pass
    over this same command (profiles/rNN_pmc_fetch.csv, already x2-corrected for gfx950's 64-B
    tally of 128-B requests).  PMC counters cannot be read from inside the process, so this is the
    profile's number, not a live one; None when no pass is committed."""
    import csv
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_fetch_fp8.csv" if quant else "r*_pmc_fetch.csv")))
    if not files:
        return None, None
    tot = n = 0
    with open(files[-1]) as f:
        for row in csv.DictReader(f):
            if kernel in row["kernel"] and row["counter"] == "FETCH_SIZE":
                tot += float(row["avg_bytes_x2_corrected"]) * int(row["launches"])
                n += int(row["launches"])
    return (int(tot / n), os.path.basename(files[-1])) if n else (None, None)


def cpu_baseline(cfg, bs, ctx, dtype, quant=None):
    """Oracle decode step on the host cores: time `nl` layers + the LM head and scale to the full
    depth (every layer does identical work).  quant = "w8a8_fp8": the oracle's fp8 W8A8 linears (per-token
    quantisation, fp8 x fp8 products in fp32, scales), the reference's non-CUDA apply_fp8_linear branch."""
    from oracle.llama_ref import OracleLlama
    from oracle import ops_ref

    nl = 2
    torch.manual_seed(0)
    small = dict(cfg)
    small["num_hidden_layers"] = nl
    from sglang_custom_amd.model_loader.synthetic import llama_param_shapes

    w = {}
    for name, shape in llama_param_shapes(small).items():
        w[name] = (torch.randn(shape, dtype=torch.bfloat16) * 0.02).to(dtype) if len(shape) > 1 \
            else torch.ones(shape, dtype=dtype)
    m = OracleLlama(small, w, dtype, max_total_tokens=bs * (ctx + 16), max_reqs=bs, context_len=ctx + 24,
                    fp8_w8a8=quant == "w8a8_fp8")
    # fill the cache as if `ctx` tokens per request were already there (contents are irrelevant
    # to the timing): allocate slots and table entries without running a prefill
    m.req_pool_indices = torch.arange(bs, dtype=torch.int64)
    m.seq_lens = torch.full((bs,), ctx, dtype=torch.int64)
    slots = m._alloc(bs * ctx).view(bs, ctx)
    m.req_to_token[:bs, :ctx] = slots.to(torch.int32)
    for c in m.k_cache + m.v_cache:
        c.normal_(0, 1)
    ids = torch.randint(0, 10000, (bs,))
    threads = torch.get_num_threads()
    import statistics

    # whole steps of the nl-layer model, then the LM head alone; median of `reps` each
    reps = 7
    m.decode(ids)  # warm-up
    t_steps, t_heads = [], []
    for _ in range(reps):
        t0 = time.perf_counter()
        m.decode(ids)
        t_steps.append(time.perf_counter() - t0)
    h = torch.randn(bs, cfg["hidden_size"]).to(dtype)
    m._logits(h)
    for _ in range(reps):
        t0 = time.perf_counter()
        m._logits(h)
        t_heads.append(time.perf_counter() - t0)
    t_head = statistics.median(t_heads)
    L = cfg["num_hidden_layers"]

    def scaled(t_step):
        return max(t_step - t_head, 1e-9) / nl * L + t_head

    full = scaled(statistics.median(t_steps))
    lo, hi = bs / scaled(max(t_steps)), bs / scaled(min(t_steps))
    return dict(value=round(bs / full, 2), unit="tok/s", cores=threads, kind="port",
                spread=[round(lo, 2), round(hi, 2)],
                sample=f"oracle {'fp8-W8A8 ' if quant else ''}decode step bs={bs} ctx={ctx}: {nl} of {L} layers + LM head, median of {reps} runs "
                       f"({(statistics.median(t_steps) - t_head) / nl * 1e3:.0f} ms/layer, LM head {t_head * 1e3:.0f} ms) on "
                       f"{threads} threads, scaled to {L} layers; spread = slowest..fastest run")


def metric_name(args, tp, simulated):
    if simulated:
        return "per-GPU kernel time of one simulated TP shard (diagnostic)"
    names = {"llama-3-8b": "Llama-3-8B", "llama-3-70b": "Llama-3-70B", "llama-3.2-1b": "Llama-3.2-1B"}
    dt = ({"bfloat16": "bf16", "float16": "f16", "float32": "f32"}[args.dtype] + (" fp8-W8A8" if args.quantization else "")
          + (" fp8-KV-cache" if args.kv_cache_dtype != "auto" else ""))
    return (f"decode output tok/s ({names.get(args.model, args.model)} {dt} TP={tp}, bs={args.batch_size}, "
            f"{args.input_len}-in) + p50 TTFT")


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"[bench] {world} rank(s) running (WORLD_SIZE) but --gpus {args.gpus}: refusing to mislabel the run",
              file=sys.stderr)
        sys.exit(2)
    n_gpus = world
    import torch.distributed as dist

    from sglang_custom_amd import bench_one_batch as bob
    from sglang_custom_amd.configs.model_config import ModelConfig
    from sglang_custom_amd.model_executor.model_runner import ModelRunner
    from sglang_custom_amd.model_loader.synthetic import hf_config_dict
    from sglang_custom_amd.server_args import ServerArgs

    if args.dry_run:
        dist.init_process_group("gloo", world_size=world, rank=rank)
        dist.barrier()
        alive = torch.ones(1)
        dist.all_reduce(alive)
        dist.barrier()
        if rank == 0:
            tp = world if args.parallelism == "tp" else 1
            print(json.dumps({"metric": metric_name(args, tp, False), "value": None, "unit": "tok/s", "n_gpus": n_gpus,
                              "ranks_alive": int(alive.item()), "steps": args.steps, "warmup": args.warmup,
                              "scaling": "strong" if tp > 1 else "weak", "dry_run": True,
                              "config": {"parallelism": f"tp{tp}" if tp > 1 else (f"dp{world}" if world > 1 else "tp1")}}),
                  flush=True)
        dist.destroy_process_group()
        return
    share = bool(args.ranks_share_gpu and world > 1)
    if share:
        if args.parallelism != "tp":
            sys.exit("--ranks-share-gpu rehearses the tensor-parallel branch only")
        local_rank = 0          # every rank on the one GPU of the box
    torch.cuda.set_device(local_rank)
    tp = world if (args.parallelism == "tp" and world > 1) else 1
    sim = args.simulate_tp if (args.simulate_tp > 1 and world == 1) else 0
    if sim:
        tp = sim
    if world > 1 and tp == 1:
        dist.init_process_group("nccl", world_size=world, rank=rank,
                                device_id=torch.device(f"cuda:{local_rank}"))
    cfg = hf_config_dict(args.model, args.dtype)
    if args.prefill_gemm:
        from sglang_custom_amd.layers.quantization import unquant

        unquant.PREFILL_GEMM = args.prefill_gemm
    bs, in_len = args.batch_size, args.input_len
    out_len = args.warmup + args.steps + 1
    sa = ServerArgs(dtype=args.dtype, max_total_tokens=bs * (in_len + out_len + 8) + 64,
                    max_running_requests=max(bs, 8) + 8, context_length=in_len + out_len + 16,
                    disable_cuda_graph=args.no_graph, cuda_graph_bs=[bs], tp_size=tp,
                    quantization=args.quantization, random_seed=rank if tp == 1 else 0,
                    simulate_tp_shard=bool(sim), kv_cache_dtype=args.kv_cache_dtype, kv_layout=args.kv_layout,
                    disable_custom_all_reduce=args.custom_all_reduce == "off",
                    custom_all_reduce_self_test=args.custom_all_reduce == "auto",
                    # a failed graph capture (collectives inside a capture can fail on a machine this build has not
                    # seen) means eager launches on the SAME runner, decided by all ranks together
                    graph_capture_fallback=True,
                    # two ranks on one device: RCCL refuses, so the process group is gloo (device tensors staged
                    # through the host) and the one-launch collectives carry the decode step
                    dist_backend="gloo" if share else None)
    mc = ModelConfig(cfg, dtype=args.dtype, context_length=in_len + out_len + 16)
    mr = ModelRunner(mc, sa, gpu_id=local_rank, tp_rank=rank if tp > 1 else 0, tp_size=tp)
    if mr.graph_capture_error and rank == 0:
        print(f"[bench] HIP-graph capture failed ({mr.graph_capture_error}); running eager", file=sys.stderr)
    graph_ok = mr.graph_runner is not None

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- TTFT: prefill latency of the whole batch, p50 over a few repeats ----
    reqs = bob.prepare_synthetic_inputs_for_latency_test(bs, in_len, seed=0)
    ttfts = []
    for r in range(args.ttft_repeats + 1):
        bob.clear_pools(mr)
        for q in reqs:
            q.req_pool_idx = None
            q.already_computed = 0
            q.cached_tokens = 0
        barrier()
        t0 = time.perf_counter()
        next_ids, _, batch = bob.extend(reqs, mr)
        torch.cuda.synchronize()
        if r > 0:  # first run is the warm-up
            ttfts.append(time.perf_counter() - t0)
    ttft = sorted(ttfts)[len(ttfts) // 2]

    # ---- decode: W warm-up + K timed steps ----
    for _ in range(args.warmup):
        next_ids, _ = bob.decode(next_ids, batch, mr)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        next_ids, _ = bob.decode(next_ids, batch, mr)
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device="cpu" if (world > 1 and dist.get_backend() == "gloo") else "cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    from sglang_custom_amd.distributed.parallel_state import check_collectives

    check_collectives()     # a custom collective that lost a peer filled its output with NaN: the run is void
    ms_per_step = elapsed * 1e3 / args.steps
    replicas = world if tp == 1 else 1
    value = bs * replicas * args.steps / elapsed

    dtype_bytes = 2 if args.dtype != "float32" else 4
    wbytes, _ = weight_bytes_per_step(cfg, tp, dtype_bytes, args.quantization)
    mean_ctx = in_len + args.warmup + (args.steps + 1) / 2.0
    kv_bytes = 1 if args.kv_cache_dtype == "fp8_e4m3" else dtype_bytes
    if "kv_lora_rank" in cfg:   # one latent row per token and layer, shared by all heads and ranks
        kv_tok = (cfg["kv_lora_rank"] + cfg["qk_rope_head_dim"]) * cfg["num_hidden_layers"] * kv_bytes
    else:
        kv_tok = 2 * max(1, cfg["num_key_value_heads"] // tp) * cfg["head_dim"] * cfg["num_hidden_layers"] * kv_bytes
    bytes_step = wbytes + bs * mean_ctx * kv_tok + bs * kv_tok
    step_gbs = bytes_step / (ms_per_step * 1e-3) / 1e9

    out = {
        "metric": metric_name(args, tp, bool(sim)),
        "value": round(value, 1), "unit": "tok/s", "n_gpus": n_gpus, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak" if tp == 1 else "strong", "vs_baseline": None,
        "dtype": {"bfloat16": "bf16", "float16": "f16", "float32": "f32"}[args.dtype]
                 + ("+fp8-w8a8" if args.quantization else "") + ("+fp8-kv" if args.kv_cache_dtype != "auto" else ""),
        "data": "synthetic (random-init weights of the named shape, np.random prompts seed 0)",
        "config": {"workload": f"{args.model} {args.dtype} decode, static batch {bs}, {in_len}-token prompts, "
                               f"greedy, context {in_len + args.warmup}..{in_len + args.warmup + args.steps}",
                   "global_batch": bs * replicas, "seq_len": in_len,
                   "parallelism": (f"ONE shard of tp{tp}, collectives stubbed (not a job-level number)" if sim else
                                   f"tp{tp} REHEARSAL: all {world} ranks share GPU 0 (not a multi-GPU number)" if share else
                                   f"tp{tp}" if tp > 1 else (f"dp{world}" if world > 1 else "tp1")),
                   "hip_graph": bool(graph_ok)},
        "ttft_p50_ms": round(ttft * 1e3, 2),
        "prefill_tok_s": round(bs * in_len / ttft, 1),
        "prefill_gemm": args.prefill_gemm or os.environ.get("MI355_PREFILL_GEMM", "tiled"),
        "step_bytes_algorithmic": int(bytes_step),
        "step_hbm_gbs": round(step_gbs, 1),
        "step_hbm_frac_of_8TBs": round(step_gbs / HBM_PEAK_GBS, 4),
    }
    if tp > 1 and not sim:
        # exchange steps of one decode step on every rank: 2 per layer (o_proj, down_proj) + the
        # embedding, each a sum over ranks of a [bs, hidden] tensor; + the all-gather of the logits shards
        n_ar = 2 * cfg["num_hidden_layers"] + 1
        from sglang_custom_amd.distributed.parallel_state import custom_all_reduce_state, get_custom_all_reduce

        ca = get_custom_all_reduce()
        out["collectives"] = {"backend": "custom one-launch xGMI kernels (RCCL above their size limit)"
                              if ca is not None and not ca.disabled else "rccl",
                              "custom_all_reduce": custom_all_reduce_state(),
                              "ranks": world, "all_reduces_per_step": n_ar,
                              "all_reduce_message_bytes": bs * cfg["hidden_size"] * dtype_bytes,
                              "all_gather_bytes_per_rank": bs * (cfg["vocab_size"] // tp) * dtype_bytes}
    mla = "kv_lora_rank" in cfg   # the per-kernel roofline and the CPU port below are written for the Llama layer
    if rank == 0 and not args.no_roofline and not mla:
        # rank 0's shard: the projection kernels launch no collective, so measuring on one rank is safe
        k = measure_dominant_kernel(mr, bs, cfg, tp, dtype_bytes, args.quantization)
        traffic, traffic_src = committed_pmc_traffic(k["kernel"], args.quantization) if tp == 1 else (None, None)
        out["roofline"] = {"bound": "hbm", "achieved": round(k["achieved_gbs"], 1), "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": round(k["achieved_gbs"] / HBM_PEAK_GBS, 4),
                           "traffic": traffic, "traffic_source": traffic_src, "kernel": k["kernel"],
                           "avg_duration_us": round(k["avg_duration_us"], 2),
                           "bytes_per_launch": int(k["bytes_per_launch"]),
                           "launches_per_step": k["launches_per_step"]}
    if rank == 0 and not args.no_cpu_baseline and world == 1 and not mla:
        del mr
        torch.cuda.empty_cache()
        out["cpu_baseline"] = cpu_baseline(cfg, bs, in_len + args.warmup + args.steps // 2,
                                           torch.bfloat16 if args.dtype == "bfloat16" else torch.float32,
                                           args.quantization)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1 and dist.is_initialized():
        barrier()   # rank 0 may still have been measuring: nobody tears the group down early
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
