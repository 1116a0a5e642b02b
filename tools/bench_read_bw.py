"""Dev tool: what a plain streaming read reaches for a launch geometry (workgroups x loads in flight)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_custom_amd import _lib
_lib.use_dev_library()
lib = _lib.load()
nbytes = int(sys.argv[1]) if len(sys.argv) > 1 else 64 * 4096 * 576 * 2
bufs = [torch.randint(0, 255, (nbytes,), dtype=torch.uint8, device='cuda') for _ in range(3)]
sink = torch.zeros(4, dtype=torch.int32, device='cuda')
st = torch.cuda.current_stream().cuda_stream
for grid in (256, 512, 1024, 2048, 4096):
    for unroll in (4, 8, 16):
        for i in range(3): lib.mi355_debug_read_bw(bufs[i].data_ptr(), nbytes, grid, unroll, sink.data_ptr(), st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(12): lib.mi355_debug_read_bw(bufs[i % 3].data_ptr(), nbytes, grid, unroll, sink.data_ptr(), st)
        e1.record(); torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / 12 * 1e3
        print(f"grid={grid:5d} loads in flight per lane={unroll:2d}: {t:7.1f} us  {nbytes / t / 1e6:5.2f} TB/s", flush=True)
