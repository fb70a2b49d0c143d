#!/bin/bash
# Developer tool: an experimental liblsq_hip.so with ONE instantiation of the single-launch quantizer (seconds to build):
#   scripts/exp_build.sh <tag> <U> <VEC> [extra -D flags]   ->  ml-quant_amd/lib_exp/<tag>/liblsq_hip.so
# (the other objects come from ml-quant_amd/lib_dbg: build that first with EXTRA=-DLSQ_PHASE_CLOCKS)
set -e
cd "$(dirname "$0")/.."
tag=$1; U=$2; VEC=$3; shift 3
out=ml-quant_amd/lib_exp/$tag; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Iinclude -Wno-unused-function -DLSQ_PHASE_CLOCKS \
  -DLSQ_DEV_U=$U -DLSQ_DEV_VEC=$VEC "$@" -c ml-quant_amd/csrc/lsq_act_fused.hip -o $out/lsq_act_fused.o
objs=$(ls ml-quant_amd/lib_dbg/*.o | grep -v lsq_act_fused.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/liblsq_hip.so $objs $out/lsq_act_fused.o
echo built $out
