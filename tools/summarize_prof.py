"""Turn rocprofv3's rocpd databases (kernel-trace / --pmc passes of bench.py) into the small CSV
summaries kept under profiles/.

  python tools/summarize_prof.py gpurun_out/prof_r01 profiles/r01
writes  profiles/r01_kernel_stats.csv   per-kernel calls / total / average / min / max (ns), % of GPU time
        profiles/r01_decode_step.csv    per-kernel launches and time inside one replayed decode step
        profiles/r01_pmc_fetch.csv      per-kernel FETCH_SIZE (raw and x2-corrected bytes per launch)
"""
import csv
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(?:mi355::)?([A-Za-z0-9_:]+)", name)
    return name if len(name) < 90 else (name[:87] + "...")


def kernel_stats(db, out_csv, step_csv):
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end, grid_x, grid_y, workgroup_x from kernels order by start").fetchall()
    agg = defaultdict(list)
    for name, s, e, *_ in rows:
        agg[name].append(e - s)
    total = sum(sum(v) for v in agg.values())
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct_gpu_time"])
        for name, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([short(name), len(v), sum(v), round(sum(v) / len(v)), min(v), max(v), round(100 * sum(v) / total, 2)])
    # one decode step = the kernels between two consecutive argmax launches near the end of the run
    idx = [i for i, r in enumerate(rows) if "argmax" in r[0]]
    if len(idx) > 3 and step_csv:
        a, b = idx[-3] + 1, idx[-2] + 1
        step = rows[a:b]
        per = defaultdict(list)
        for name, s, e, gx, gy, wx in step:
            per[(name, gx // max(wx, 1), gy)].append(e - s)
        span = step[-1][2] - step[0][1]
        busy = sum(e - s for _, s, e, *_ in step)
        with open(step_csv, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "workgroups_x", "grid_y", "launches", "total_ns", "avg_ns"])
            for (name, gx, gy), v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
                w.writerow([short(name), gx, gy, len(v), sum(v), round(sum(v) / len(v))])
            w.writerow(["# step span ns (first start -> last end)", "", "", len(step), span, ""])
            w.writerow(["# sum of kernel durations ns", "", "", "", busy, ""])
            w.writerow(["# idle between kernels ns", "", "", "", span - busy, ""])


def pmc(db, out_csv):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, counter_name, value from counters_collection").fetchall()
    agg = defaultdict(list)
    for k, cn, v in rows:
        agg[(k, cn)].append(v)
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "launches", "avg_value_raw", "avg_bytes_x2_corrected", "max_value_raw"])
        for (k, cn), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            avg = sum(v) / len(v)
            # FETCH_SIZE is in KiB of 64-B requests; gfx950 tallies 128-B requests of a wide
            # coalesced read at 64 B, so bytes = value * 1024 * 2
            w.writerow([short(k), cn, len(v), round(avg, 1), round(avg * 1024 * 2), max(v)])


if __name__ == "__main__":
    src, dst = sys.argv[1], sys.argv[2]
    kernel_stats(f"{src}/trace/bench_results.db", f"{dst}_kernel_stats.csv", f"{dst}_decode_step.csv")
    try:
        pmc(f"{src}/pmc_fetch/bench_results.db", f"{dst}_pmc_fetch.csv")
    except Exception as e:  # noqa: BLE001
        print("no pmc pass:", e)
