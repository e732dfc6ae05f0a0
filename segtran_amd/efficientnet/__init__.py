from .model import EfficientNet, MBConvBlock  # noqa: F401
