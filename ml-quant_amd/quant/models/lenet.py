"""LeNet-5 with a quantized second convolution.

Same constructor, sub-module names and forward as the reference's ``quant/models/lenet.py``
(:21-94): first and last layers full precision, ``conv2`` a 5x5 ``QuantConv2d`` over 20
input channels (not a multiple of 64, no padding) fed by a non-affine batch norm.
"""

from typing import Callable, Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from quant.binary.binary_conv import QuantConv2d


class QLeNet5(nn.Module):
    """conv5-relu-bn-pool, bn-quantconv5-relu-pool, fc-relu-fc, log-softmax."""

    def __init__(self, loss_fn: Callable[..., torch.Tensor], conv1_filters: int = 20, conv2_filters: int = 50,
                 output_classes: int = 10, x_quant: str = 'fp', w_quant: str = 'fp',
                 clamp: Optional[Dict] = None, moving_average_mode: str = 'off',
                 moving_average_momentum: float = 0.99) -> None:
        super().__init__()
        setattr(self, 'loss_fn', loss_fn)
        self.conv1_filters, self.conv2_filters = conv1_filters, conv2_filters
        self.output_classes = output_classes
        self.x_quant, self.w_quant = x_quant, w_quant

        self.conv1 = nn.Conv2d(1, conv1_filters, 5, stride=1)
        self.bn_conv1 = nn.BatchNorm2d(conv1_filters, eps=1e-4, momentum=0.1, affine=False)
        self.conv2 = QuantConv2d(x_quant, w_quant, conv1_filters, conv2_filters, 5, clamp,
                                 moving_average_mode, moving_average_momentum, stride=1)
        self.bn_conv2 = nn.BatchNorm2d(conv1_filters, eps=1e-4, momentum=0.1, affine=False)
        hidden = conv2_filters * output_classes
        self.fc1 = nn.Linear(conv2_filters * 4 * 4, hidden)
        self.fc2 = nn.Linear(hidden, output_classes)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = F.max_pool2d(self.bn_conv1(F.relu(self.conv1(x))), kernel_size=2, stride=2)
        # bn -> quantized conv -> relu: one fused call (on the HIP path the batch norm is folded into the
        # quantizer's read and the ReLU into the conv epilogue; elsewhere it is the plain composition)
        x = F.max_pool2d(self.conv2.fused_forward(x, self.bn_conv2, relu=True), kernel_size=2, stride=2)
        x = F.relu(self.fc1(x.reshape(-1, self.conv2_filters * 4 * 4)))
        return F.log_softmax(self.fc2(x), dim=1)
