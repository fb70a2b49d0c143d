"""MI355X-native least-squares binary quantization, behind the reference's module API.

Module paths and public names mirror apple/ml-quant's ``quant`` package for the quantized
forward path so model code and yaml configs written for the reference
run unchanged; CUDA (ROCm) tensors in eval mode are served by hand-written gfx950 kernels
through ``quant._hip`` (C ABI in ``include/lsq_hip.h``).
"""

__version__ = '0.1.0'
