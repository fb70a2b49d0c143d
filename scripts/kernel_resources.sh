#!/bin/bash
# Developer tool: registers, scratch and LDS of the kernels in an object file of ml-quant_amd/lib (from the code object's notes).
# usage: scripts/kernel_resources.sh ml-quant_amd/lib/lsq_act_fused.o [name filter]
obj=$1; filt=${2:-.}
B=/opt/rocm/lib/llvm/bin
tmp=$(mktemp -d)
$B/llvm-objcopy --dump-section .hip_fatbin=$tmp/fat.bin $obj 2>/dev/null
$B/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$tmp/fat.bin --output=$tmp/dev.o --unbundle
$B/llvm-readelf --notes $tmp/dev.o | awk '
/\.name:/ {name=$2} /\.vgpr_count:/ {v=$2} /\.agpr_count:/ {ag=$2} /\.sgpr_count:/ {s=$2} /\.private_segment_fixed_size:/ {p=$2} /\.group_segment_fixed_size:/ {l=$2}
/\.wavefront_size:/ {print name, "vgpr", v, "agpr", ag, "sgpr", s, "scratch", p, "lds", l}' | c++filt | grep -E "$filt" | cut -c1-160
rm -rf $tmp
