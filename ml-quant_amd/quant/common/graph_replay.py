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
        side = torch.cuda.Stream(device=example.device)
        side.wait_stream(torch.cuda.current_stream(example.device))
        with torch.no_grad(), torch.cuda.stream(side):
            for _ in range(max(1, warmup)):                 # allocates plane workspaces, packs the weights, folds the batch norms
                self.model(self.static_input)
        torch.cuda.current_stream(example.device).wait_stream(side)
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_output = self.model(self.static_input)

    def replay(self) -> torch.Tensor:
        """Run the captured forward on whatever ``static_input`` holds."""
        self.graph.replay()
        return self.static_output

    def __call__(self, x: Optional[torch.Tensor] = None) -> torch.Tensor:
        if x is not None and x.data_ptr() != self.static_input.data_ptr():
            self.static_input.copy_(x)
        return self.replay()
