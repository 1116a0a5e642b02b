#!/bin/bash
# Development A/B: builds of the library that differ only in -DGT_ABL=<n> of gemm_tiled.hip
# (needs sglang_custom_amd/build_dev/*.o from `python -m sglang_custom_amd.build --dev`).
set -e
cd "$(dirname "$0")/../sglang_custom_amd"
mkdir -p build_var
FLAGS="-O3 --offload-arch=gfx950 -fPIC -std=c++17 -fno-gpu-rdc -Wno-unused-result -mllvm -amdgpu-mfma-vgpr-form -DMI355_DEV_BUILD"
for v in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS $GT_EXTRA -DGT_ABL=$v -c csrc/gemm_tiled.hip -o build_var/gemm_tiled_$v.o &
done
wait
OBJS=$(ls build_dev/*.o | grep -v gemm_tiled)
for v in "$@"; do
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o build_var/libgt_$v.so $OBJS build_var/gemm_tiled_$v.o
done
ls -la build_var/*.so
