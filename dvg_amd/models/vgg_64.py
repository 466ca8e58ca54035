"""models.vgg_64 counterpart: 64x64 VGG encoder / decoder (reference vgg_64.py:17-159)."""
from .backbones import VggDecoder, VggEncoder, VggGaussianEncoder, vgg_layer  # noqa: F401


class encoder(VggEncoder):
    RES = 64


class decoder(VggDecoder):
    RES = 64


class gaussian_encoder(VggGaussianEncoder):
    RES = 64
