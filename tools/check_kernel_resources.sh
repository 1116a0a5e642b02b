#!/bin/bash
# Lists every gfx950 kernel of the library that uses private (scratch) memory or spills registers:
# both are performance bugs in this code base (arrays that fell out of registers).  Build container only.
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
cd "$tmp"
bad=0
allow="__no_kernel_is_exempt__"
for f in "$root"/sglang_custom_amd/csrc/*.hip; do
  b=$(basename "$f" .hip)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -I"$root/include" -c "$f" -o "$b.o" -save-temps 2>/dev/null
  s="$b-hip-amdgcn-amd-amdhsa-gfx950.s"
  n=$(grep -E "^\s+\.(name|private_segment_fixed_size|vgpr_spill_count):" "$s" | paste - - - | grep -v "$allow" | awk '$4 != 0 || $6 != 0' | wc -l)
  echo "$b: $(grep -c '^\s*\.name:' "$s") kernels, $n with private memory or spills"
  grep -E "^\s+\.(name|private_segment_fixed_size|vgpr_spill_count):" "$s" | paste - - - | awk '$4 != 0 || $6 != 0 {print "   ", $2, "private", $4, "spills", $6}' | cut -c1-200
  bad=$((bad + n))
done
rm -rf "$tmp"
exit $((bad > 0))
