#!/bin/bash
# Runs on the GPU box: the bench lines quoted beside the headline one (other batch sizes, fp8 modes, simulated TP shards), each saved as
# its own record.  Usage: tools/bench_matrix.sh <tag>     -> gpurun_out/matrix_<tag>/*.json
set -o pipefail
tag=${1:-r02}
out=gpurun_out/matrix_$tag
mkdir -p $out
run() {  # name, args...
  name=$1; shift
  timeout -k 10 400 python3 bench.py --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err || { echo "$name failed"; tail -3 $out/$name.err; return 1; }
  cut -c1-200 $out/$name.json
}
run bench_bs1 --batch-size 1 && run bench_bs8 --batch-size 8 && run bench_bs32 --batch-size 32 && run bench_bs64 --batch-size 64 && \
run bench_fp8_w8a8 --quantization w8a8_fp8 && run bench_fp8_kv --kv-cache-dtype fp8_e4m3 && \
run bench_70b_tp8_shard_bs1 --model llama-3-70b --simulate-tp 8 --batch-size 1 --input-len 4096 --steps 32 --ttft-repeats 1 && \
run bench_70b_tp8_shard_bs16 --model llama-3-70b --simulate-tp 8 --batch-size 16 --input-len 4096 --steps 32 --ttft-repeats 1 && \
run bench_70b_tp8_shard_bs64 --model llama-3-70b --simulate-tp 8 --batch-size 64 --input-len 4096 --steps 32 --ttft-repeats 1
