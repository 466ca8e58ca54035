"""Alias of dvg_amd.models.vgg_64 under the reference's module path."""
from dvg_amd.models.vgg_64 import *  # noqa: F401,F403
from dvg_amd.models import vgg_64 as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
