for r in 1 2 3; do for t in old base st1; do
  if [ $t = base ]; then L=$PWD/ml-quant_amd/lib/liblsq_hip.so; else L=$PWD/ml-quant_amd/lib_exp/$t/liblsq_hip.so; fi
  LSQ_HIP_LIB=$L python bench.py --no-configs --cpu-sample 0 --min-seconds 2 --detail gpurun_out/det_${t}_$r.json 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$t', round(d['value']), 'single', round(d['single_stream']['value']), 'quant', round(r['quantizer']['ms_per_step'],4), 'xnor', round(r['xnor_conv']['ms_per_step'],4))"
done; done
