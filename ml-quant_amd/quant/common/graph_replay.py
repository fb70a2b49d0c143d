"""HIP-graph replay of an eval forward: for launch-bound configurations (small images, small batches) the forty-odd
kernel launches of a quantized ResNet cost more host time than GPU time; captured once in a HIP graph
(``torch.cuda.CUDAGraph`` -- the C-ABI launches go to torch's current stream and are captured like any other kernel),
the whole forward is ONE launch.  Nothing of the reference corresponds to this (it runs eager PyTorch); it is the
serving-side counterpart of ``evaluate`` for fixed input shapes.

    fwd = GraphedForward(model, example_input)      # warms up (workspaces, packed weights), captures
    logits = fwd(x)                                  # copies x into the static input, replays, returns the static output
"""

from typing import Optional

import torch


class GraphedForward:
    """Eval-mode ``model`` captured for inputs of ``example.shape`` on ``example.device``."""

    def __init__(self, model: torch.nn.Module, example: torch.Tensor, warmup: int = 3) -> None:
        if not example.is_cuda:
            raise ValueError('graph capture needs a CUDA (ROCm) tensor')
        self.model = model.eval()
        self.static_input = example.clone()
        self.graph = torch.cuda.CUDAGraph()
        # Warm-up and capture run on a stream of THIS object: the kernels' workspaces (solver / sweep scratch, plane
        # buffers) are cached per launch stream, and torch's default capture stream is shared by every capture of the
        # process -- two graphs captured on it would share scratch allocated in the first graph's pool and race on it
        # when replayed on different streams.
        self.stream = torch.cuda.Stream(device=example.device)
        self.stream.wait_stream(torch.cuda.current_stream(example.device))
        with torch.no_grad(), torch.cuda.stream(self.stream):
            for _ in range(max(1, warmup)):                 # allocates plane workspaces, packs the weights, folds the batch norms
                self.model(self.static_input)
        torch.cuda.current_stream(example.device).wait_stream(self.stream)
        # (the stem's fp16-domain report is not collected inside a capture: read it now, and let the model switch to the
        # bf16 split BEFORE its kernels are frozen into the graph)
        from quant import _hip as _h
        if _h.stem_overflow_check(example.device):
            with torch.no_grad(), torch.cuda.stream(self.stream):
                self.model(self.static_input)
            torch.cuda.current_stream(example.device).wait_stream(self.stream)
        with torch.no_grad(), torch.cuda.graph(self.graph, stream=self.stream):
            self.static_output = self.model(self.static_input)
        # the graph replays raw addresses: everything the captured kernels touch outside the graph's own pool -- packed
        # weights, plane buffers and scratch allocated during the warm-up -- stays referenced for the graph's lifetime,
        # whatever the modules' and the binding's caches evict later
        from quant import _hip
        key = (self.static_input.device.index, self.stream.cuda_stream)
        from quant.binary import chain
        self._keepalive = [_hip._ws_cache.get(key), _hip._sweep_ws_cache.get(key), chain._arenas.get(key)]
        for m in self.model.modules():
            cache = getattr(m, '_hip_cache', None)
            if isinstance(cache, dict):
                self._keepalive.append(list(cache.values()))

    def replay(self) -> torch.Tensor:
        """Run the captured forward on whatever ``static_input`` holds."""
        self.graph.replay()
        return self.static_output

    def __call__(self, x: Optional[torch.Tensor] = None) -> torch.Tensor:
        if x is not None and x.data_ptr() != self.static_input.data_ptr():
            self.static_input.copy_(x)
        return self.replay()
