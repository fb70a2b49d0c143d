"""The limits of the free-running GPU parity assertions are DERIVED, not calibrated: tests/golden/free_limits.json is
what tests/golden/make_free_limits.py computes on the CPU from the oracle and the reference fixtures (the deviation the
exact argmin alone explains, optimal.py:151).  This file re-derives them and checks the committed numbers."""

import importlib.util
import json
import os

import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _maker():
    spec = importlib.util.spec_from_file_location('make_free_limits', os.path.join(GOLDEN, 'make_free_limits.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _committed():
    with open(os.path.join(GOLDEN, 'free_limits.json')) as f:
        return json.load(f)


def test_single_layer_limits_are_what_the_oracle_gives():
    mk, doc = _maker(), _committed()
    fresh = mk.conv_layers()
    for key in ('conv_fixture', 'conv_lenet', 'conv_geometry'):
        assert fresh[key]['tiebreak'] == pytest.approx(doc['limits'][key]['tiebreak'], rel=1e-3, abs=1e-9), key
        assert doc['limits'][key]['limit'] == pytest.approx(1.05 * doc['limits'][key]['tiebreak'] + 1e-4, rel=1e-9), key
    # the exact argmin does move the reference's outputs (this is not a vacuous bound), by a few per cent of max|y| at most
    assert 1e-3 < doc['limits']['conv_fixture']['tiebreak'] < 5e-2


def test_block_and_network_limits_are_what_the_oracle_gives():
    mk, doc = _maker(), _committed()
    blk = mk.block_case()
    for key in ('block', 'block_cos'):
        assert blk[key]['tiebreak'] == pytest.approx(doc['limits'][key]['tiebreak'], rel=1e-3, abs=1e-12), key
        want = 1.05 * doc['limits'][key]['tiebreak'] + 2.0 * doc['limits'][key]['sensitivity']
        assert doc['limits'][key]['limit'] == pytest.approx(max(want, 1e-9) if 'cos' in key else want, rel=1e-9), key
    nets = doc['networks']
    modular = ('imagenet_ls1w_ls2a', 'imagenet_ls1w_lsTa', 'imagenet_ls1w_gf2a', 'cifar100_ls1')
    assert doc['limits']['resnet_logits']['limit'] == pytest.approx(
        max(1.05 * nets[t]['tiebreak'] + 2.0 * nets[t]['sensitivity'] for t in modular), rel=1e-9)
    # schemes without a search (gf-2, ls-1, fp) have nothing to tie-break: the exact-argmin oracle IS the reference
    for tag in ('imagenet_ls1w_gf2a', 'cifar100_ls1', 'imagenet_ls1w_fpa'):
        assert nets[tag]['tiebreak'] == 0.0, tag
    # amplified arithmetic noise stays orders of magnitude below the tie-break term on these inputs
    assert all(v['sensitivity'] < 1e-4 for v in nets.values())


def test_one_network_rederived():
    mk, doc = _maker(), _committed()
    mk.NETS = {'imagenet_ls1w_ls2a': (2, 3, 64, 64)}
    fresh = mk.net_cases()['imagenet_ls1w_ls2a']
    want = doc['networks']['imagenet_ls1w_ls2a']
    assert fresh['tiebreak'] == pytest.approx(want['tiebreak'], rel=1e-3)
    assert fresh['tiebreak_cos'] == pytest.approx(want['tiebreak_cos'], rel=1e-2)
