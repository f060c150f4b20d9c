from .modules import Conv2d, BatchNorm2d, Linear, LayerNorm, conv_bn  # noqa: F401
