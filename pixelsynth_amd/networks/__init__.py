"""Dense networks around the hot path: depth Unet and refinement decoder, reference-compatible."""
from .architectures import ResNetDecoder, ResNet_Block, Unet, get_decoder  # noqa: F401
