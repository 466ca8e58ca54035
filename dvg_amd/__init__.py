"""dvg_amd — MI355X-native (gfx950) implementation of the DVG frame-prediction hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed);
every hot op is a hand-written HIP kernel in dvg_amd/csrc/libdvg_hip.so reached through
the C ABI of include/dvg_hip.h.  There is no CPU or eager-PyTorch fallback.
"""
__version__ = "0.1.0"
