#!/usr/bin/env python3
"""Developer tool: per-phase cycle stamps of lsq_act_quant's workgroup 0 (needs a library built with
`make -C ml-quant_amd/csrc EXTRA=-DLSQ_PHASE_CLOCKS`)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
import torch
from quant import _hip
lib = _hip.lib()
names = {10: 'solve start', 11: 'slots loaded', 13: 'roles set', 4: 'gather done', 5: 'wave path done', 6: 'block path done', 12: 'argmin done'}
for c, h in ((64, 56), (128, 28), (256, 14), (512, 7)):
    n = 256
    x = torch.randn(n, c, h, h, device='cuda')
    g = _hip.make_geom(n, c, h, h, c, 3, 3, (1, 1), (1, 1), (1, 1), 1)
    planes = torch.zeros(2 * _hip.act_plane_words(g), dtype=torch.int64, device='cuda')
    scales = torch.empty((2, n), device='cuda')
    for _ in range(3):
        _hip.act_quant(x, g, 2, 2, 3, 3.0, planes, scales)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 32)()
    lib.lsq_debug_read_clocks(buf)
    t0 = buf[10]
    print(f'C={c} H={h}: ' + ', '.join(f'{names[i]}={buf[i] - t0}' for i in names if buf[i]))
    print(f'      hist sweep: sweep + reductions {buf[8] - buf[15]}, level-1 scan {buf[14] - buf[8]}, slot records {buf[16] - buf[14]}')
    print(f'      wave 0, last slot: hist {buf[21] - buf[20]}, sub-bin scan {buf[22] - buf[21]}, flagged sub-bins {buf[23] - buf[22]}  (segment {buf[25]} keys, {buf[26]} flagged sub-bins)')
    ws = (ctypes.c_longlong * 64)()
    lib.lsq_debug_read_wave_stats(ws)
    print('      wave path per wave (cycles/slots/flagged sub-bins/ranked keys): ' +
          ' '.join(f'{ws[4 * w]}/{ws[4 * w + 1]}/{ws[4 * w + 2]}/{ws[4 * w + 3]}' for w in range(16)))
    bt = (ctypes.c_longlong * 2048)()
    lib.lsq_debug_read_block_times(bt)
    import numpy as np
    t = np.array(bt[:2 * n]).reshape(n, 2).astype(np.float64)
    start, dur = (t[:, 0] - t[:, 0].min()) / 100.0, (t[:, 1] - t[:, 0]) / 100.0      # 100 MHz -> us
    end = (t[:, 1] - t[:, 0].min()) / 100.0
    print(f'      solve per row (us): start spread {start.max():.1f}; duration min {dur.min():.1f} median {np.median(dur):.1f} '
          f'p90 {np.percentile(dur, 90):.1f} max {dur.max():.1f}; last row ends at {end.max():.1f}')
    st = (ctypes.c_longlong * 4096)()
    lib.lsq_debug_read_sweep_times(st)
    sw = np.array(st[:]).reshape(2, 1024, 2)[:, :n, :].astype(np.float64) / 100.0
    t0 = sw[0, :, 0].min()
    print(f'      timeline (us from the first histogram-sweep workgroup): hist sweep {sw[0, :, 0].min() - t0:.1f}..{sw[0, :, 1].max() - t0:.1f} | '
          f'solve {t[:, 0].min() / 100.0 - t0:.1f}..{t[:, 1].max() / 100.0 - t0:.1f} | plane-1 sweep {sw[1, :, 0].min() - t0:.1f}..{sw[1, :, 1].max() - t0:.1f}')
