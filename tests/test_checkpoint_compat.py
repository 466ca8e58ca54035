"""CPU: a checkpoint pickled by the REFERENCE (whole nn.Module objects, train.py:380-383) must unpickle into our
classes through the top-level `models` alias package, with an identical attribute tree.  Needs the reference tree
(this container only; skipped on the GPU box)."""
import importlib
import io
import os
import sys

import pytest
import torch

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")


def _ref_import(name):
    saved = {k: v for k, v in sys.modules.items() if k == "models" or k.startswith("models.")}
    for k in saved:
        del sys.modules[k]
    old_path = list(sys.path)
    sys.path[:] = [REF] + [p for p in old_path if os.path.abspath(p or ".") not in (ROOT, os.getcwd())]
    try:
        mod = importlib.import_module(f"models.{name}")
        assert mod.__file__.startswith(REF)
        return mod, saved, old_path
    except Exception:
        sys.path[:] = old_path
        raise


def _restore(saved, old_path):
    sys.path[:] = old_path
    for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
        del sys.modules[k]
    sys.modules.update(saved)


@pytest.mark.parametrize("name", ["dcgan_64", "vgg_64"])
def test_reference_pickle_loads_into_our_classes(name):
    torch.manual_seed(0)
    mod, saved, old_path = _ref_import(name)
    try:
        enc, dec = mod.encoder(90, 1), mod.decoder(90, 1)
        ref_keys = (list(enc.state_dict().keys()), list(dec.state_dict().keys()))
        ref_w = enc.state_dict()["c2.main.0.weight" if name == "dcgan_64" else "c2.0.main.0.weight"].clone()
        buf = io.BytesIO()
        torch.save({"encoder": enc, "decoder": dec}, buf)   # pickles by class path models.<name>.encoder
    finally:
        _restore(saved, old_path)
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    for k in [k for k in sys.modules if k == "models" or k.startswith("models.")]:
        del sys.modules[k]
    buf.seek(0)
    ck = torch.load(buf, weights_only=False)   # resolves models.<name> -> our alias package
    ours = importlib.import_module(f"dvg_amd.models.{name}")
    assert isinstance(ck["encoder"], ours.encoder) and isinstance(ck["decoder"], ours.decoder)
    assert list(ck["encoder"].state_dict().keys()) == ref_keys[0]
    assert list(ck["decoder"].state_dict().keys()) == ref_keys[1]
    key = "c2.main.0.weight" if name == "dcgan_64" else "c2.0.main.0.weight"
    assert torch.equal(ck["encoder"].state_dict()[key], ref_w)
    # the unpickled object has OUR forward (HIP path): calling it on CPU must refuse, not silently run torch.nn
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ck["encoder"](torch.zeros(1, 1, 64, 64))
    # layer wrappers are our classes too (the decoder walks isinstance(layer, vgg_layer))
    if name == "vgg_64":
        assert isinstance(ck["decoder"].upc2[0], ours.vgg_layer)
