"""Deterministic, platform-independent input/parameter generation shared by the fixture
generator (``make_fixtures.py``, run once in the build container against the reference)
and by the tests (run anywhere).  Uses numpy's frozen legacy ``RandomState`` streams, so
fixtures only need to store *expected outputs*.
"""

import zlib

import numpy as np
import torch


def _seed(tag: str, seed: int) -> int:
    return (zlib.crc32(tag.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF


def normal(tag: str, shape, seed: int = 0, scale: float = 1.0) -> torch.Tensor:
    rs = np.random.RandomState(_seed(tag, seed))
    return torch.from_numpy((rs.standard_normal(size=tuple(shape)) * scale).astype(np.float32))


def uniform(tag: str, shape, lo: float = 0.0, hi: float = 1.0, seed: int = 0) -> torch.Tensor:
    rs = np.random.RandomState(_seed(tag, seed))
    return torch.from_numpy((lo + (hi - lo) * rs.random_sample(size=tuple(shape))).astype(np.float32))


def fill_module(module: torch.nn.Module, seed: int = 0) -> None:
    """Overwrite every parameter and BatchNorm statistic of ``module`` deterministically.

    Conv/linear weights ~ N(0, 1/fan_in); biases ~ N(0, 0.1); BN gamma in [0.5, 1.5],
    beta ~ N(0, 0.2), running_mean ~ N(0, 0.3), running_var in [0.5, 1.5]; PReLU slope 0.25.
    Quantizer scale buffers (``v1`` ...) and moving-average state are left untouched.
    """
    with torch.no_grad():
        for name, p in module.named_parameters():
            if name.endswith('.weight') and p.dim() >= 2:
                fan_in = int(np.prod(p.shape[1:]))
                p.copy_(normal(name, p.shape, seed, scale=fan_in ** -0.5))
            elif name.endswith('.weight') and p.dim() == 1 and p.numel() > 1:  # BN gamma
                p.copy_(uniform(name, p.shape, 0.5, 1.5, seed))
            elif name.endswith('.weight'):                    # PReLU single slope
                p.fill_(0.25)
            elif name.endswith('.bias'):
                owner = module.get_submodule(name.rsplit('.', 1)[0])
                sc = 0.2 if isinstance(owner, torch.nn.modules.batchnorm._BatchNorm) else 0.1
                p.copy_(normal(name, p.shape, seed, scale=sc))
        for name, b in module.named_buffers():
            if name.endswith('running_mean'):
                b.copy_(normal(name, b.shape, seed, scale=0.3))
            elif name.endswith('running_var'):
                b.copy_(uniform(name, b.shape, 0.5, 1.5, seed))
