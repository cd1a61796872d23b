"""Dense networks around the hot path: depth Unet and refinement decoder, reference-compatible."""
from .architectures import ResNetDecoder, ResNet_Block, Unet, get_decoder  # noqa: F401
from .discriminators import MultiscaleDiscriminator, NLayerDiscriminator, define_D  # noqa: F401
from .resnet import ResNet18, resnet18  # noqa: F401
