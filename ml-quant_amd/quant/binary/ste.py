"""{-1,+1} sign and its straight-through estimator.

Interface of the reference's ``quant/binary/ste.py``: ``binary_sign`` (:16-18), ``STESign``
(:21-66) and ``binarize`` (:70).  sign(+0) = sign(-0) = +1.
"""

import torch


def binary_sign(x: torch.Tensor) -> torch.Tensor:
    """+1 where x >= 0 (including -0.0), -1 where x < 0; same dtype as x."""
    one = torch.ones((), dtype=x.dtype, device=x.device)
    return torch.where(x < 0, -one, one)


class STESign(torch.autograd.Function):
    """Sign in the forward pass, clipped identity (|x| <= 1) in the backward pass."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return binary_sign(x)

    @staticmethod
    def backward(ctx, grad_output):
        (x,) = ctx.saved_tensors
        return grad_output * (x.abs() <= 1).to(grad_output.dtype)


binarize = STESign.apply
