"""Alias of dvg_amd.models.lstm under the reference's module path."""
from dvg_amd.models.lstm import *  # noqa: F401,F403
from dvg_amd.models import lstm as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
