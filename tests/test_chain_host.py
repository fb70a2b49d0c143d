"""Host side of the chained 1-bit layers (quant/binary/chain.py): the hand-over record and the per-forward accumulators.
The kernels behind it are covered on the GPU (tests/test_gpu_round4.py)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'ml-quant_amd')]
from quant.binary import chain  # noqa: E402


def test_record_rides_on_the_tensor_object_and_dies_with_an_in_place_edit():
    y = torch.zeros(2, 3)
    rec = chain.PreQuant(consumer='conv', pre_bn=None, planes=None, units=None, shape=(2, 3), stream=0)
    chain.attach(y, rec, keep=('planes', 'units'))
    assert chain.pending(y) is rec
    assert chain.pending(y.clone()) is None and chain.pending(y + 0) is None       # another tensor: nothing to consume
    assert chain.pending(y.detach()) is None                                        # (the consumer looks BEFORE it detaches)
    y.add_(1.0)                                                                     # the values moved: the record is void
    assert chain.pending(y) is None
    z = torch.zeros(2, 3)
    chain.attach(z, rec, keep=None)
    chain.ENABLED = False
    try:
        assert chain.pending(z) is None
    finally:
        chain.ENABLED = True
    assert chain.pending(z) is rec


def test_accumulators_of_a_forward_share_one_zeroed_arena():
    dev = torch.device('cpu')
    # outside a scope (or on another device than the scope's): every accumulator is its own zeroed tensor
    a = chain.accumulator(4, dev)
    assert a.dtype == torch.int64 and a.shape == (4,) and int(a.abs().sum()) == 0
    with chain.scope(torch.device('meta')):
        b = chain.accumulator(4, dev)
        assert b.data_ptr() != a.data_ptr() and int(b.abs().sum()) == 0
    # nested scopes restore the outer one
    with chain.scope(torch.device('meta')):
        with chain.scope(torch.device('meta')):
            pass
        assert chain._state.arena is not None
    assert getattr(chain._state, 'arena', None) is None


def test_stream_pipeline_is_a_plain_call_on_the_cpu():
    from quant.common.stream_pipeline import StreamPipeline, eval_streams
    model = torch.nn.Linear(4, 3)
    pipe = StreamPipeline(model, 'cpu', 2)
    xs = [torch.randn(5, 4, generator=torch.Generator().manual_seed(i)) for i in range(4)]
    assert pipe.depth == 1 and eval_streams('cpu') == 1 and eval_streams('cuda', sharded=True) == 2 and eval_streams('cuda') == 2
    with torch.no_grad():
        got = list(pipe.map(xs))
        assert all(torch.equal(g, model(x)) for g, x in zip(got, xs))
        assert torch.equal(pipe.submit(xs[0]).result(), model(xs[0]))
