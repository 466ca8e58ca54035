"""models.dcgan_64 counterpart (reference dcgan_64.py:28-88; decoder output Tanh)."""
from ..ops import ACT_TANH
from .backbones import DcganDecoder, DcganEncoder, dcgan_conv, dcgan_upconv  # noqa: F401


class encoder(DcganEncoder):
    RES = 64


class decoder(DcganDecoder):
    RES = 64
    FINAL_ACT = ACT_TANH
