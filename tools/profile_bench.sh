#!/bin/bash
# Runs on the GPU box: kernel-trace/stats pass and (separately) one PMC pass over bench.py.
# Usage: tools/profile_bench.sh <tag> [bench.py arguments ...]     outputs under gpurun_out/prof_<tag>/
set -o pipefail
tag=${1:-r01}
shift
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $out/trace -o bench -- python3 bench.py "$@" > $out/bench_trace.log 2>&1 || { tail -5 $out/bench_trace.log; exit 1; }
grep "\"metric\"" $out/bench_trace.log > $out/bench_line.json
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch -o bench -- python3 bench.py --steps 4 --warmup 2 --no-roofline --no-cpu-baseline "$@" > $out/bench_pmc.log 2>&1 || { tail -5 $out/bench_pmc.log; exit 1; }
find $out -name "*.csv" | head -20
