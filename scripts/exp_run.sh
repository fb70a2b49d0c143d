LSQ_HIP_LIB=$PWD/ml-quant_amd/lib_exp/sq/liblsq_hip.so python -m pytest tests/ -q -m gpu -k "stem" 2>&1 | tail -2
LSQ_HIP_LIB=$PWD/ml-quant_amd/lib_exp/sqc/liblsq_hip.so python scripts/stem_clocks.py 2>&1 | grep "split 22"
for r in 1 2 3; do echo "== base"; python scripts/stem_time.py 2>&1 | grep -v amdgpu | tail -1; echo "== sq"; LSQ_HIP_LIB=$PWD/ml-quant_amd/lib_exp/sq/liblsq_hip.so python scripts/stem_time.py 2>&1 | grep -v amdgpu | tail -1; done
