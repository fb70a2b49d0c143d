#!/usr/bin/env python3
"""Developer tool: phase timeline of the sign-weight fast-path kernel (a library built with -DLSQ_SIGNW_CLOCKS).

    LSQ_HIP_LIB=.../lib_tune/CLK/liblsq_hip.so python scripts/signw_clocks.py C H O stride

Per workgroup: hardware ids (XCC, SE, CU, SIMD wave slot) and s_memtime stamps (100 MHz constant clock on gfx950? -- the
script prints raw deltas) at kernel start, around the conversion / MFMA phases of the first chunks, before and after the
epilogue.  Prints the phases of the workgroups that share a CU side by side.
"""
import ctypes
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch  # noqa: E402
from quant import _hip  # noqa: E402

c, h, o, stride = (int(v) for v in sys.argv[1:5])
n, dev = 256, 'cuda:0'
x = torch.randn(n, c, h, h, device=dev)
w = torch.randn(o, c, 3, 3, device=dev)
g = _hip.make_geom(n, c, h, h, o, 3, 3, (stride, stride), (1, 1), (1, 1), 1)
wsc = w.abs().mean(dim=(1, 2, 3)).view(1, -1).contiguous()
wbits, _ = _hip.pack_weight(w, g, wsc)
wprep = _hip.signw_prepare_weight(wbits, 1, g)
ho, wo = _hip.out_hw(g)
y = torch.empty((n, o, ho, wo), device=dev)
bias = torch.zeros(o, device=dev)
for _ in range(3):
    _hip.signw_conv2d(x, 2.0, wbits, wsc, bias, g, y, wprep=wprep)
torch.cuda.synchronize()
nwg = 1 << 16
buf = torch.zeros((nwg, 64), dtype=torch.int64, device=dev)
lib = _hip.lib()
lib.lsq_debug_signw_clocks.argtypes = [ctypes.c_void_p]
assert lib.lsq_debug_signw_clocks(buf.data_ptr()) == 0
_hip.signw_conv2d(x, 2.0, wbits, wsc, bias, g, y, wprep=wprep)
torch.cuda.synchronize()
lib.lsq_debug_signw_clocks(None)
b = buf.cpu().numpy()
used = [i for i in range(nwg) if b[i, 2] != 0]
print(f'{len(used)} workgroups')
t0 = min(b[i, 2] for i in used)
cus = defaultdict(list)
for i in used:
    hw = int(b[i, 0])
    wave, simd, cu, sh, se = hw & 15, (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
    cus[(int(b[i, 1]), se, sh, cu)].append((int(b[i, 2]), i, wave, simd))
shown = 0
for key in sorted(cus):
    wgs = sorted(cus[key])
    if shown < 6:
        print(f'XCC {key[0]} SE {key[1]} SH {key[2]} CU {key[3]}: {len(wgs)} workgroups')
        for start, i, wave, simd in wgs:
            r = b[i]
            ph = []
            for cc in range(6):
                cs, ms, me = r[4 + 3 * cc], r[5 + 3 * cc], r[6 + 3 * cc]
                if cs:
                    ph.append(f'[c{cc}: conv@{cs - t0} {ms - cs} | mfma {me - ms}]')
            print(f'   wg {i:5d} slot {wave} simd {simd} start {start - t0:8d} tables {r[60] - start:6d} barrier {r[61] - start:6d} loop_end {r[3] - t0:8d} end {r[63] - t0:8d}  ' + ' '.join(ph))
        shown += 1
per = defaultdict(int)
for key, wgs in cus.items():
    per[len(wgs)] += 1
print('workgroups per CU histogram:', dict(per))
import numpy as np  # noqa: E402
conv = np.array([[b[i, 5 + 3 * cc] - b[i, 4 + 3 * cc] for cc in range(1, 4)] for i in used if b[i, 13] != 0])
mfma = np.array([[b[i, 6 + 3 * cc] - b[i, 5 + 3 * cc] for cc in range(1, 4)] for i in used if b[i, 13] != 0])
gap = np.array([[b[i, 4 + 3 * (cc + 1)] - b[i, 6 + 3 * cc] for cc in range(1, 3)] for i in used if b[i, 16] != 0])
print('median ticks: convert+barrier', np.median(conv), ' mfma phase', np.median(mfma), ' end barrier', np.median(gap),
      ' epilogue', np.median([b[i, 63] - b[i, 3] for i in used]), ' total', np.median([b[i, 63] - b[i, 2] for i in used]))
