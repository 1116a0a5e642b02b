#!/bin/bash
# Runs on the GPU box: KV-split sweep of the fused decode attention launch at a tensor-parallel rank's small-batch shapes
# (one kv head, 8 query heads, ~4100 keys; qkv cut 11 ways).  Usage: tools/sweep_small_batch_splits.sh > log
set -o pipefail
for bs in 1 16; do
  for ks in 2 4 6 8 10 12 16 24 32 48 64; do
    echo "bs=$bs"
    QKV_SPLITS=11 KV_SPLITS=$ks timeout -k 10 120 python tools/bench_attn_fused.py $bs 8 1 4104 2>&1 | grep -v amdgpu.ids || exit 1
  done
done
