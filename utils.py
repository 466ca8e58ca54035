"""`import utils` (train.py:9, generate_frames.py:9): the path-relevant helpers of the reference's
utils.py, restated in dvg_amd/utils.py.  Image / GIF writers and SSIM are out of scope."""
from dvg_amd.utils import init_weights, normalize_data  # noqa: F401
