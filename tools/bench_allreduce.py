"""Latency of the one-launch all-reduce with the ranks sharing ONE GPU (what a one-GPU box can
measure: launch + barrier protocol + copies through uncached memory; no xGMI in the path).
usage: python tools/bench_allreduce.py [world]"""
import multiprocessing as mp
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(rank, world, port):
    import torch
    import torch.distributed as dist

    from sglang_custom_amd.distributed.device_communicators.custom_all_reduce import CustomAllreduce

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    ca = CustomAllreduce(dist.group.WORLD, 0)
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
            if rank == 0:
                print(f"world {world} bytes {count * 2:8d} algo {algo}: {e0.elapsed_time(e1) * 1000 / 200:7.2f} us", flush=True)
            del g
    ca.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=main, args=(r, world, 29533)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        if p.is_alive():
            p.kill()
    sys.exit(max(p.exitcode or 0 for p in procs))
