"""models.dcgan_128 counterpart (reference dcgan_128.py:28-94; decoder output Sigmoid)."""
from ..ops import ACT_SIGMOID
from .backbones import DcganDecoder, DcganEncoder, dcgan_conv, dcgan_upconv  # noqa: F401


class encoder(DcganEncoder):
    RES = 128


class decoder(DcganDecoder):
    RES = 128
    FINAL_ACT = ACT_SIGMOID
