#!/usr/bin/env python3
"""Generate the golden vectors of tests/golden/*.npz by RUNNING THE REFERENCE's own modules.

Run in the build container only (needs /root/reference, CPU is enough):

    python tests/golden/make_golden.py

The reference's `models/{vgg_64,vgg_128,dcgan_64,dcgan_128,lstm}.py` are imported unmodified
from /root/reference (only `torch.Tensor.cuda` is shimmed to identity in THIS harness because
lstm.py:61-62 hard-calls `.cuda()`).  Weights and inputs come from oracle/params.py (numpy
PCG64, seed-determined), so only the reference OUTPUTS are stored: the fixtures are data, never
reference source.  gp_models.py cannot be imported (needs gpytorch) -> no GP fixture: GP parity
is unpinned (see oracle/dvg_oracle.py).
"""
import importlib
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import params  # noqa: E402


def ref_import(name):
    """Import `models.<name>` from the reference tree without shadowing our top-level `models`."""
    saved = {k: v for k, v in sys.modules.items() if k == "models" or k.startswith("models.")}
    for k in saved:
        del sys.modules[k]
    # the reference's models/ has no __init__.py (namespace package): a regular `models`
    # package anywhere on sys.path would win, so our repo root must be hidden meanwhile
    old_path = list(sys.path)
    sys.path[:] = [REF] + [p for p in old_path if os.path.abspath(p or ".") not in (ROOT, os.getcwd())]
    try:
        mod = importlib.import_module(f"models.{name}")
        assert mod.__file__.startswith(REF), mod.__file__
    finally:
        sys.path[:] = old_path
        for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
            del sys.modules[k]
        sys.modules.update(saved)
    return mod


def summarize(t: torch.Tensor, k: int = 64) -> np.ndarray:
    """[sum, abs-sum, sum of squares, k strided samples]: cheap fingerprint of a big skip tensor."""
    f = t.detach().double().reshape(-1)
    idx = torch.linspace(0, f.numel() - 1, k).long()
    return torch.cat([torch.stack([f.sum(), f.abs().sum(), (f * f).sum()]), f[idx]]).numpy()


def run_backbone(family: str, res: int, nc: int, batch: int, training: bool, seed: int, out: dict, tag: str):
    torch.manual_seed(0)
    mod = ref_import(f"{family}_{res}")
    enc, dec = mod.encoder(90, nc), mod.decoder(90, nc)
    # our attribute tree must be key-for-key identical
    import dvg_amd.models as ours_pkg  # noqa: F401
    ours = importlib.import_module(f"dvg_amd.models.{family}_{res}")
    assert list(ours.encoder(90, nc).state_dict().keys()) == list(enc.state_dict().keys()), "encoder keys differ"
    assert list(ours.decoder(90, nc).state_dict().keys()) == list(dec.state_dict().keys()), "decoder keys differ"
    esd = params.fill_state_dict(enc.state_dict(), seed)
    dsd = params.fill_state_dict(dec.state_dict(), seed + 1, params.decoder_transposed_keys(dec.state_dict(), family))
    enc.load_state_dict(esd)
    dec.load_state_dict(dsd)
    enc.train(training)
    dec.train(training)
    x = params.frames(seed + 2, batch, nc, res)
    with torch.no_grad():
        h, skips = enc(x)
        vec = params.normal(seed + 3, batch, 90, scale=0.5).tanh()  # decoder fed an independent latent
        y = dec([vec, skips])
        y_h = dec([h, skips])
    out[f"{tag}/h"] = h.numpy()
    for i, s in enumerate(skips):
        out[f"{tag}/skip{i}"] = summarize(s)
    out[f"{tag}/y"] = y.numpy()
    out[f"{tag}/y_h"] = y_h.numpy()
    if training:  # BatchNorm side effects (running statistics after ONE encoder + TWO decoder calls)
        esd2, dsd2 = enc.state_dict(), dec.state_dict()
        for k in ("c1.0.main.1.running_mean", "c1.0.main.1.running_var", "c5.1.running_mean", "c5.1.running_var"):
            if k in esd2:
                out[f"{tag}/enc/{k}"] = esd2[k].numpy()
        for k in ("upc1.1.running_mean", "upc1.1.running_var", "upc2.0.main.1.running_var", "upc2.main.1.running_var"):
            if k in dsd2:
                out[f"{tag}/dec/{k}"] = dsd2[k].numpy()
    print(f"{tag}: h {tuple(h.shape)} |h|max {h.abs().max():.3f} y mean {y.mean():.4f} std {y.std():.4f}")


def run_lstm(out: dict):
    torch.Tensor.cuda = lambda self, *a, **k: self  # harness-only shim for lstm.py:61-62
    mod = ref_import("lstm")
    import dvg_amd.models.lstm as ours
    for cls, tag, seed in (("lstm", "lstm", 300), ("gaussian_lstm", "gaussian_lstm", 310)):
        B = 5
        net = getattr(mod, cls)(90, 90, 256, 2, B)
        assert list(getattr(ours, cls)(90, 90, 256, 2, B).state_dict().keys()) == list(net.state_dict().keys())
        sd = params.fill_state_dict(net.state_dict(), seed)
        net.load_state_dict(sd)
        net.hidden = net.init_hidden()
        ys = []
        with torch.no_grad():
            for t in range(3):
                x = params.normal(seed + 10 + t, B, 90, scale=0.5)
                if cls == "lstm":
                    ys.append(net(x).numpy())
                else:
                    torch.manual_seed(1234 + t)  # the reference draws eps from the global RNG (lstm.py:163)
                    z, mu, logvar = net(x)
                    torch.manual_seed(1234 + t)
                    eps = torch.randn(B, 90)
                    ys.append(np.stack([z.numpy(), mu.numpy(), logvar.numpy(), eps.numpy()]))
        out[f"{tag}/y"] = np.stack(ys)
        out[f"{tag}/h1"] = net.hidden[1][0].detach().numpy()
        out[f"{tag}/c1"] = net.hidden[1][1].detach().numpy()
        print(f"{tag}: y std {np.stack(ys).std():.4f}")


def run_gaussian_encoder(out: dict):
    """vgg_64.gaussian_encoder (vgg_64.py:108-159): trunk + mu / logvar heads + reparameterisation.  The reference draws eps
    with `logvar.data.new(size).normal_()` from the global RNG (:150-153); it is recovered by replaying the seed."""
    mod = ref_import("vgg_64")
    import dvg_amd.models.vgg_64 as ours
    B, seed = 3, 180
    net = mod.gaussian_encoder(90, 24, 1)
    assert list(ours.gaussian_encoder(90, 24, 1).state_dict().keys()) == list(net.state_dict().keys())
    sd = params.fill_state_dict(net.state_dict(), seed)
    net.load_state_dict(sd)
    net.eval()
    x = params.frames(seed + 2, B, 1, 64)
    with torch.no_grad():
        torch.manual_seed(4321)
        z, mu, logvar, skips = net(x)
        torch.manual_seed(4321)
        eps = torch.empty(B, 24).normal_()
    assert torch.equal(z, eps * torch.exp(0.5 * logvar) + mu) or torch.allclose(z, eps * torch.exp(0.5 * logvar) + mu,
                                                                                 rtol=0, atol=1e-6)
    out["gaussian_encoder/zmle"] = np.stack([z.numpy(), mu.numpy(), logvar.numpy(), eps.numpy()])
    for i, sk in enumerate(skips):
        out[f"gaussian_encoder/skip{i}"] = summarize(sk)
    print(f"gaussian_encoder: mu std {mu.std():.4f} logvar std {logvar.std():.4f}")


# ---- gradients: what the REFERENCE's own `.backward()` (train.py:240) produces ------------------------------------
GRAD_BATCH = 16   # >= 16 so that single LeakyReLU-kink flips stop dominating batch-statistics BatchNorm gradients


def run_backbone_grads(family: str, seed: int, out: dict, tag: str):
    """encoder -> decoder([h, skips]) in TRAIN mode at B=16, loss = sum(y * Gy) + sum(h * Gh) with seeded upstream
    gradients, `.backward()` on the reference modules; per-parameter gradient fingerprints (sum, |sum|, sum of squares,
    64 strided samples) are stored, plus the outputs."""
    torch.manual_seed(0)
    mod = ref_import(f"{family}_64")
    enc, dec = mod.encoder(90, 1), mod.decoder(90, 1)
    esd = params.fill_state_dict(enc.state_dict(), seed)
    dsd = params.fill_state_dict(dec.state_dict(), seed + 1, params.decoder_transposed_keys(dec.state_dict(), family))
    enc.load_state_dict(esd)
    dec.load_state_dict(dsd)
    enc.train(), dec.train()
    x = params.frames(seed + 2, GRAD_BATCH, 1, 64)
    gy = params.normal(seed + 4, GRAD_BATCH, 1, 64, 64)
    gh = params.normal(seed + 5, GRAD_BATCH, 90)
    h, skips = enc(x)
    y = dec([h, skips])
    ((y * gy).sum() + (h * gh).sum()).backward()
    out[f"{tag}/h"] = h.detach().numpy()
    out[f"{tag}/y"] = summarize(y)
    for name, net in (("enc", enc), ("dec", dec)):
        for k, p_ in net.named_parameters():
            out[f"{tag}/{name}/{k}"] = summarize(p_.grad)
    print(f"{tag}: {sum(1 for k in out if k.startswith(tag + '/enc/') or k.startswith(tag + '/dec/'))} gradient fingerprints")


def run_lstm_grads(out: dict):
    """4-step BPTT through lstm.lstm (lstm.py:42-72) at B=16: loss = sum_t sum(y_t * G_t)."""
    torch.Tensor.cuda = lambda self, *a, **k: self
    mod = ref_import("lstm")
    B, seed = GRAD_BATCH, 320
    net = mod.lstm(90, 90, 256, 2, B)
    net.load_state_dict(params.fill_state_dict(net.state_dict(), seed))
    net.hidden = net.init_hidden()
    xs = [params.normal(seed + 10 + t, B, 90, scale=0.5).requires_grad_(True) for t in range(4)]
    loss = sum((net(xs[t]) * params.normal(seed + 20 + t, B, 90)).sum() for t in range(4))
    loss.backward()
    out["lstm_grad/loss"] = np.array([float(loss)])
    for k, p_ in net.named_parameters():
        out[f"lstm_grad/{k}"] = summarize(p_.grad)
    for t in range(4):
        out[f"lstm_grad/x{t}"] = xs[t].grad.numpy()
    print("lstm_grad: loss", float(loss))


def write_reference_checkpoint():
    """A checkpoint exactly as the reference's train.py:380-388 writes it - WHOLE pickled nn.Module objects of the reference's
    own classes (models.dcgan_64.encoder / decoder, models.lstm.lstm: train.py:75 hard-codes dcgan_64) plus the GP /
    likelihood state_dicts under gpytorch-0.3.x key names and an `opt` Namespace with train.py:17-46's fields - with every
    tensor ZEROED so that the gzip'ed file is a few KB: the test fills the weights from oracle/params.py seeds after
    loading.  The fixture is data (class paths + zero tensors); unpickling it on the GPU box resolves the class paths to
    this repository's alias package."""
    import argparse
    import gzip
    import io
    torch.Tensor.cuda = lambda self, *a, **k: self
    mod, lmod = ref_import("dcgan_64"), ref_import("lstm")
    enc, dec = mod.encoder(90, 1), mod.decoder(90, 1)
    fp = lmod.lstm(90, 90, 256, 2, 50)
    with torch.no_grad():
        for m in (enc, dec, fp):
            for t in list(m.parameters()) + list(m.buffers()):
                t.zero_()
        fp.hidden = fp.init_hidden()
    gsd, lik = params.gp_state(0)
    gsd = OrderedDict((k, torch.zeros_like(v)) for k, v in gsd.items())
    lik = OrderedDict((k, torch.zeros_like(v)) for k, v in lik.items())
    opt = argparse.Namespace(lr=0.002, beta1=0.9, batch_size=50, log_dir='logs', model_dir='', name='', output_path='.',
                             data_root='path/to/data/', optimizer='adam', niter=601, seed=1, epoch_size=300, image_width=64,
                             channels=1, dataset='kth', n_past=5, ft=True, n_future=10, n_eval=15, rnn_size=256,
                             predictor_rnn_layers=2, z_dim=10, g_dim=90, model='dcgan', data_threads=5,
                             last_frame_skip=False)
    # pickling by reference needs `models.<name>` importable as the REFERENCE's modules at dump time
    saved = {k: v for k, v in sys.modules.items() if k == "models" or k.startswith("models.")}
    for k in saved:
        del sys.modules[k]
    import types
    pkg = types.ModuleType("models")
    pkg.__path__ = [os.path.join(REF, "models")]
    sys.modules.update({"models": pkg, "models.dcgan_64": mod, "models.lstm": lmod})
    try:
        buf = io.BytesIO()
        torch.save({'encoder': enc, 'decoder': dec, 'frame_predictor': fp, 'likelihood': lik, 'gp_layer': gsd,
                    'gp_layer_optimizer': {}, 'opt': opt}, buf)
    finally:
        for k in ("models", "models.dcgan_64", "models.lstm"):
            sys.modules.pop(k, None)
        sys.modules.update(saved)
    path = os.path.join(HERE, "reference_checkpoint_zeroed.pth.gz")
    with gzip.open(path, "wb", compresslevel=9) as f:
        f.write(buf.getvalue())
    print("wrote", path, os.path.getsize(path) // 1024, "KiB (", len(buf.getvalue()) // 2**20, "MiB raw )")


def main():
    out = OrderedDict()
    run_backbone("vgg", 64, 1, 2, False, 100, out, "vgg_64/eval")
    run_backbone("vgg", 64, 1, 4, True, 110, out, "vgg_64/train")
    run_backbone("dcgan", 64, 1, 2, False, 120, out, "dcgan_64/eval")
    run_backbone("dcgan", 64, 1, 4, True, 130, out, "dcgan_64/train")
    run_backbone("vgg", 64, 3, 2, False, 140, out, "vgg_64_nc3/eval")
    run_backbone("dcgan", 64, 3, 2, False, 150, out, "dcgan_64_nc3/eval")
    run_backbone("vgg", 128, 3, 1, False, 160, out, "vgg_128/eval")
    run_backbone("dcgan", 128, 3, 2, False, 170, out, "dcgan_128/eval")
    run_lstm(out)
    run_gaussian_encoder(out)
    run_backbone_grads("dcgan", 200, out, "dcgan_64/grad")
    run_backbone_grads("vgg", 210, out, "vgg_64/grad")
    run_lstm_grads(out)
    path = os.path.join(HERE, "reference_outputs.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")
    write_reference_checkpoint()


if __name__ == "__main__":
    main()
