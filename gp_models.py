"""`from gp_models import GPRegressionLayer1` (generate_frames.py:14 imports it top-level)."""
from dvg_amd.models.gp_models import *  # noqa: F401,F403
from dvg_amd.models.gp_models import GPRegressionLayer1, GaussianLikelihood, VariationalELBO  # noqa: F401
