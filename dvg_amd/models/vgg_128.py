"""models.vgg_128 counterpart: 128x128 VGG encoder / decoder (reference vgg_128.py:16-120)."""
from .backbones import VggDecoder, VggEncoder, vgg_layer  # noqa: F401


class encoder(VggEncoder):
    RES = 128


class decoder(VggDecoder):
    RES = 128
