"""Round-4 GPU tests: the solver's chosen CANDIDATE (its sorted position, not only its value) pinned on the device."""

import numpy as np
import pytest
import torch

import detgen
from oracle import lsq_exact as E

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _hip():
    from quant import _hip
    return _hip


def _solver_rows():
    """The rows of fixture f3_solver (tests/test_oracle_golden.py builds the same)."""
    rows = {'long': detgen.normal('f3.long', (4, 25088), scale=1.0).clamp(-3, 3),
            'relu': detgen.normal('f3.relu', (4, 3000)).clamp(min=0),
            'sat': detgen.normal('f3.sat', (4, 3001)).clamp(-0.5, 0.5)}
    for n in (3, 4, 5, 7, 10, 11, 64):
        rows[f'short{n}'] = detgen.normal(f'f3.short{n}', (6, n))
    mix = detgen.uniform('f3.mix', (8, 768))
    mix[1] = 2.0
    mix[5] = -3.0
    mix[6] = detgen.uniform('f3.mix6', (768,), 1.0, 1.2)
    rows['mix'] = mix
    return rows


def _check_positions(tag, rows, ternary, skip, v1, pos, g, what):
    """v1 [R] and pos [R] from the device against the sorted sub-sample and the REFERENCE's candidate positions
    (fixture f3_solver `_pos`: optimal.py:78-82 evaluated by the reference itself)."""
    key = f'{tag}_t{int(ternary)}_s{skip}'
    raises = key + '_raises' in g
    for r in range(rows.shape[0]):
        row = rows[r].numpy()
        a = np.sort(E.abs_subsample(row, skip))
        n = a.shape[0]
        d = {}
        exact = E.solve_row(row, ternary, skip, d)
        p, v = int(pos[r]), np.float32(v1[r])
        assert v == exact, (what, key, r, v, exact)
        if not d['positions']:
            assert p == -1 and v == 0.0, (what, key, r, p)                  # no candidate: the zero padding wins
            continue
        pick = d['positions'][int(np.argmin(d['cost_sq']))]
        if pick < 0:
            assert ternary and p == n + 1, (what, key, r, p)                # the appended mean / 2 (optimal.py:114-116)
            continue
        # the device's position is the first element of the chosen key's run in the ascending sub-sample ...
        assert 0 <= p < n and a[p] == v and (p == 0 or a[p - 1] < v), (what, key, r, p, v)
        assert p == int(np.searchsorted(a, v, side='left')) and a[pick] == v, (what, key, r, p, pick)
        if raises:
            continue
        # ... and lies in one of the REFERENCE's candidate clusters: within 3 sorted positions of a position the
        # reference itself flagged (its fp32 means only add neighbours), or inside a run of equal keys that holds one
        refpos = [int(q) for q in g.np(key + '_pos')[r] if q >= 0]
        assert any(abs(p - q) <= 3 or a[q] == v for q in refpos), (what, key, r, p, sorted(refpos))


@pytest.mark.parametrize('ternary', [False, True])
@pytest.mark.parametrize('skip', [1, 3])
def test_chosen_candidate_position_on_device(golden, ternary, skip):
    """lsq_solve_rows (streaming path) and lsq_act_quant (single-launch kernel, its block path and its short key list)
    report through lsq_debug_solver_trace WHICH sorted element they chose: it must be the exact oracle's, the first of
    its run, and inside one of the reference's own candidate clusters."""
    hip = _hip()
    g = golden('f3_solver')
    for tag, rows in _solver_rows().items():
        with hip.solver_trace(rows.shape[0], DEV) as trace:
            v12, _ = hip.solve_rows(rows.to(DEV), skip, ternary)
            torch.cuda.synchronize()
            pos = trace.cpu().numpy().copy()
        _check_positions(tag, rows, ternary, skip, v12[0].cpu().numpy(), pos, g, 'solve_rows')
        m = rows.shape[1]
        if skip != 3 or m % 4 or m < 16:
            continue
        # the same rows as one-channel images through the quantizer entry point: the single-launch kernel
        x = rows.view(rows.shape[0], 1, 4, m // 4).contiguous()
        geom = hip.make_geom(x.shape[0], 1, 4, m // 4, 1, 1, 1, (1, 1), (0, 0), (1, 1), 1)
        words = hip.act_plane_words(geom)
        for mode in (0, 1, 2):
            planes = torch.zeros((2 * words,), dtype=torch.int64, device=DEV)
            scales = torch.empty((2, x.shape[0]), dtype=torch.float32, device=DEV)
            with hip.debug_switches(fused_mode=mode), hip.solver_trace(x.shape[0], DEV) as trace:
                hip.act_quant(x.to(DEV), geom, hip.SCHEME_LST if ternary else hip.SCHEME_LS2, 2, 3, -1.0, planes, scales)
                torch.cuda.synchronize()
                pos = trace.cpu().numpy().copy()
            _check_positions(tag, rows, ternary, skip, scales[0].cpu().numpy(), pos, g, f'act_quant mode {mode}')


def test_trace_is_off_by_default_and_after_the_block():
    hip = _hip()
    rows = detgen.normal('r4.trace', (3, 4096)).clamp(-3, 3)
    with hip.solver_trace(3, DEV) as trace:
        hip.solve_rows(rows.to(DEV), 3, False)
    first = trace.cpu().clone()
    assert (first >= 0).all()
    trace.fill_(-7)
    hip.solve_rows(rows.to(DEV), 3, False)
    torch.cuda.synchronize()
    assert (trace.cpu() == -7).all()                   # the hook is cleared: nothing writes to the old buffer
