"""Consecutive batches of an eval forward on alternating HIP streams.

Every kernel of the quantized forward fills the chip for most of its length -- and leaves it partly idle while its workgroups
are dispatched, while the last round of its tiles drains and across the boundary to the next launch: a quarter of the short
launches.  A second, independent batch fills those holes: with batch i on one stream and batch i + 1
on another, the ramp and the tail of one forward's kernels run under the other forward's kernels.  Measured on an MI355X
(batch 256, ``scripts/two_batches.py``): 2.54 -> 2.32 ms per batch for ls-1w / ls-2a, 2.78 -> 2.48 ms for ls-1w / fp-a; three
streams are slower than two; the logits are those of the single-stream forward bit for bit (every workspace of the binding
and of the modules is per launch stream).  Nothing of the reference corresponds to this: its ``evaluate``
(``quant/common/training.py:24-52``) runs one batch after the other on PyTorch's current stream.

    pipe = StreamPipeline(model, device)          # two streams on a GPU; a plain call on the CPU
    pending = [pipe.submit(x) for x in batches]   # returns at once: the forward is queued on the next stream
    outputs = [p.result() for p in pending]       # ordered after the forward on the CALLER's current stream
"""

from typing import List, Optional

import torch


class Pending:
    """A forward in flight: ``result()`` makes the caller's current stream wait for it and hands the output over."""

    __slots__ = ('_out', '_done', '_device')

    def __init__(self, out, done, device) -> None:
        self._out, self._done, self._device = out, done, device

    def result(self):
        if self._done is not None:
            cur = torch.cuda.current_stream(self._device)
            cur.wait_event(self._done)
            for t in (self._out if isinstance(self._out, (tuple, list)) else (self._out,)):
                if isinstance(t, torch.Tensor):
                    t.record_stream(cur)          # (allocated on the side stream: the allocator must not reuse it under the reader)
            self._done = None
        return self._out


class StreamPipeline:
    """``submit(x)`` runs ``model(x)`` on the next of ``streams`` HIP streams (round robin).  The forward is ordered after
    everything the caller's current stream had queued when ``submit`` was called (the input is ready) -- and after nothing
    that is queued later, so consuming batch i - 1 on the caller's stream does not hold up batch i + 1."""

    def __init__(self, model: torch.nn.Module, device, streams: int = 2) -> None:
        self.model = model
        self.device = torch.device(device)
        self.streams: List[torch.cuda.Stream] = []
        if self.device.type == 'cuda' and streams > 1:
            self.streams = [torch.cuda.Stream(device=self.device) for _ in range(streams)]
        self._next = 0
        self._last_done: Optional[torch.cuda.Event] = None

    @property
    def depth(self) -> int:
        """Forwards that can be in flight at once (1: no side streams, ``submit`` is a plain call)."""
        return max(1, len(self.streams))

    def submit(self, x: torch.Tensor) -> Pending:
        if not self.streams:
            return Pending(self.model(x), None, self.device)
        cur = torch.cuda.current_stream(self.device)
        side = self.streams[self._next % len(self.streams)]
        self._next += 1
        side.wait_event(cur.record_event())
        if self._next <= len(self.streams) and self._last_done is not None:
            # The first forward on EACH stream runs behind the forward submitted before it: a model that was just built,
            # loaded or trained fills its module-level caches (packed weight planes, prepared sign weights, folded batch norms:
            # keyed by tensor version, not by stream) on the first stream, and a second stream that got a host-side cache hit
            # would read those buffers with nothing ordering it after the writes.  From the second round on every cache is
            # warm and the streams run free.
            side.wait_event(self._last_done)
        with torch.cuda.stream(side):
            out = self.model(x)
            done = side.record_event()
        self._last_done = done
        if isinstance(x, torch.Tensor):
            x.record_stream(side)
        return Pending(out, done, self.device)

    def map(self, batches):
        """Outputs of ``model`` over ``batches``, in order, with up to ``depth`` forwards in flight."""
        window: List[Pending] = []
        for x in batches:
            window.append(self.submit(x))
            if len(window) >= self.depth:
                yield window.pop(0).result()
        while window:
            yield window.pop(0).result()


def eval_streams(device, sharded: bool = False) -> int:
    """Streams ``evaluate`` uses: two on a GPU -- also when the batch is sharded over ranks: the local forwards alternate
    between the two streams and the logits' all-gather stays on the caller's stream, behind the forward's event, in the same
    order on every rank (round 5; until then a sharded evaluation dropped to one stream) --, one on the CPU."""
    return 2 if torch.device(device).type == 'cuda' else 1
