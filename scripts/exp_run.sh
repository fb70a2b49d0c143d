for r in 1 2; do for t in sw sv; do echo "== $t"; LSQ_HIP_LIB=$PWD/ml-quant_amd/lib_exp/$t/liblsq_hip.so python scripts/stem_time.py 2>&1 | grep -v amdgpu | tail -1; done; done
LSQ_HIP_LIB=$PWD/ml-quant_amd/lib_exp/sv/liblsq_hip.so python -m pytest tests/test_gpu_round3.py tests/test_gpu_parity.py -q -m gpu -k "stem" 2>&1 | tail -2
LSQ_HIP_LIB=$PWD/ml-quant_amd/lib_exp/svc/liblsq_hip.so python scripts/stem_clocks.py 2>&1 | grep "split 22"
