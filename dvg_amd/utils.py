"""Host-side helpers of the path that the reference keeps in utils.py (restated, not imported:
the reference's utils.py needs skimage / removed scipy APIs)."""
from __future__ import annotations

import torch


def init_weights(m):
    """utils.py:304-311: Conv*/Linear ~ N(0, 0.02), bias 0; BatchNorm weight ~ N(1, 0.02), bias 0.
    Keys off class-name substrings exactly like the reference (so wrappers such as `vgg_layer` do
    not match and nn.LSTMCell keeps its default init)."""
    classname = m.__class__.__name__
    if classname.find("Conv") != -1 or classname.find("Linear") != -1:
        m.weight.data.normal_(0.0, 0.02)
        m.bias.data.fill_(0)
    elif classname.find("BatchNorm") != -1:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)


def normalize_data(opt, dtype, sequence):
    """utils.py:86-95: (B,T,H,W,C) -> list of T tensors (B,C,H,W) on the device.
    Accepts both `x` and `(x, targets)` batches (the reference unpacks a pair, which only KTH/UCF
    provide — SURVEY.md §5 quirks); returns (list, targets-or-None)."""
    targets = None
    if isinstance(sequence, (tuple, list)):
        sequence, targets = sequence
    to_gpu = dtype is None or getattr(dtype, "is_cuda", False)
    if to_gpu and torch.is_tensor(sequence) and sequence.dtype == torch.float32:
        # One host-to-device copy of the batch as it is, the three transposes as ONE device pass: the frames are the T
        # contiguous slices of a (T,B,C,H,W) buffer.  Same values as the host-side form below (pure data movement), which
        # spent 21 ms on the host per (16,12,64,64,3) batch in strided copies and 12 pageable uploads.
        seq = sequence.cuda(non_blocking=True).transpose(0, 1).transpose(3, 4).transpose(2, 3).contiguous()
        frames = [seq[t] for t in range(seq.shape[0])]
    else:
        device = torch.device("cuda") if dtype is None else None
        seq = sequence.transpose(0, 1).transpose(3, 4).transpose(2, 3)  # (T,B,C,H,W)
        if device is not None:
            frames = [seq[t].contiguous().to(device=device, dtype=torch.float32) for t in range(seq.shape[0])]
        else:
            frames = [seq[t].contiguous().type(dtype) for t in range(seq.shape[0])]
    if targets is not None and torch.is_tensor(targets) and torch.cuda.is_available():
        targets = targets.cuda()
    return frames, targets
