#!/bin/bash
# Developer tool: an experimental liblsq_hip.so with a variant of ONE kernel file: scripts/exp_file.sh <tag> <file.hip> [-D flags]
set -e
cd "$(dirname "$0")/.."
tag=$1; f=$2; shift 2
out=ml-quant_amd/lib_exp/$tag; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Iinclude -Wno-unused-function "$@" -c ml-quant_amd/csrc/$f.hip -o $out/$f.o
objs=$(ls ml-quant_amd/lib/*.o | grep -v /$f.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/liblsq_hip.so $objs $out/$f.o
echo built $out
