"""Latency of the one-launch all-reduce.
usage: python tools/bench_allreduce.py [--ranks N] [--one-gpu-per-rank]
Without --one-gpu-per-rank the ranks share cuda:0 (what a one-GPU box can measure: launch + barrier
protocol + copies through uncached memory; no xGMI in the path).  With it, rank r runs on cuda:r and
the same table is printed next to RCCL's all-reduce on the same tensors."""
import argparse
import multiprocessing as mp
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(rank, world, port, per_rank_gpu=False):
    import torch
    import torch.distributed as dist

    from sglang_custom_amd.distributed.device_communicators.custom_all_reduce import CustomAllreduce

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    dev = rank if per_rank_gpu else 0
    torch.cuda.set_device(dev)
    ca = CustomAllreduce(dist.group.WORLD, dev)
    rccl = dist.new_group(backend="nccl") if per_rank_gpu else None   # two ranks on one GPU: RCCL refuses
    for count in (4096, 8 * 4096, 64 * 4096, 256 * 4096):
        for algo in (1, 2):
            x = torch.randn(count, device="cuda").bfloat16()
            o = torch.empty_like(x)
            ca.all_reduce(x, out=o, algo=algo)
            torch.cuda.synchronize()
            dist.barrier()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(50):
                    ca.all_reduce(x, out=o, algo=algo)
            g.replay()
            torch.cuda.synchronize()
            dist.barrier()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            ca.check_status()
            line = f"world {world} bytes {count * 2:8d} algo {algo}: {e0.elapsed_time(e1) * 1000 / 200:7.2f} us"
            del g
            if rccl is not None and algo == 1:
                y = x.clone()
                for _ in range(5):
                    dist.all_reduce(y, group=rccl)
                torch.cuda.synchronize()
                dist.barrier()
                e0.record()
                for _ in range(50):
                    dist.all_reduce(y, group=rccl)
                e1.record()
                torch.cuda.synchronize()
                line += f"   rccl (eager launches): {e0.elapsed_time(e1) * 1000 / 50:7.2f} us"
            if rank == 0:
                print(line, flush=True)
    ca.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("world", nargs="?", type=int, default=None)
    ap.add_argument("--ranks", type=int, default=2)
    ap.add_argument("--one-gpu-per-rank", action="store_true")
    a = ap.parse_args()
    world = a.world or a.ranks
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=main, args=(r, world, 29533, a.one_gpu_per_rank)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        if p.is_alive():
            p.kill()
    sys.exit(max(p.exitcode or 0 for p in procs))
