#!/bin/bash
# Developer tool: an experimental liblsq_hip.so with a variant of the stem kernel: scripts/exp_stem.sh <tag> [-D flags]
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
out=ml-quant_amd/lib_exp/$tag; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Iinclude -Wno-unused-function "$@" -c ml-quant_amd/csrc/lsq_stem.hip -o $out/lsq_stem.o
objs=$(ls ml-quant_amd/lib/*.o | grep -v lsq_stem.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/liblsq_hip.so $objs $out/lsq_stem.o
scripts/kernel_resources.sh $out/lsq_stem.o "2, true"
