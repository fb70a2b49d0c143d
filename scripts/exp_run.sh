for r in 1 2; do echo "== base"; python scripts/pointwise_one.py 2>&1 | grep -v amdgpu | head -3 | cut -c1-90; echo "== pw1"; LSQ_HIP_LIB=$PWD/ml-quant_amd/lib_exp/pw1/liblsq_hip.so python scripts/pointwise_one.py 2>&1 | grep -v amdgpu | head -3 | cut -c1-90; done
LSQ_HIP_LIB=$PWD/ml-quant_amd/lib_exp/pw1/liblsq_hip.so python -m pytest tests/ -q -m gpu -k "pointwise" 2>&1 | tail -2
