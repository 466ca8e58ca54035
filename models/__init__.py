"""Top-level `models` package with the reference's import paths (`import models.dcgan_64 as model`,
train.py:75; `models.lstm`, train.py:76; `models.gp_models`, train.py:15).  Reference-pickled
checkpoints name these module paths, so they must resolve here.  Implementation: dvg_amd.models."""
