"""Model definitions built on QuantConv2d."""
