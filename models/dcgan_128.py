"""Alias of dvg_amd.models.dcgan_128 under the reference's module path."""
from dvg_amd.models.dcgan_128 import *  # noqa: F401,F403
from dvg_amd.models import dcgan_128 as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
