"""Scaled binary quantization: sign/STE, quantizers, quantizer modules and QuantConv2d."""
