"""GPU: a checkpoint in the REFERENCE's format (train.py:380-388: whole pickled modules of the reference's classes +
gpytorch-0.3.x state_dicts + opt Namespace; fixture tests/golden/reference_checkpoint_zeroed.pth.gz, written by
tests/golden/make_golden.py from the reference's own classes with zeroed tensors) must load on the GPU box, where
/root/reference does not exist, into this repository's classes - and, given the golden cases' seeded weights, reproduce
the outputs the reference's modules produced (tests/golden/reference_outputs.npz) on the HIP path."""
import gzip
import io
import os

import numpy as np
import pytest
import torch

from oracle import params
from tests.common import rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


def _load():
    raw = gzip.open(os.path.join(ROOT, "tests", "golden", "reference_checkpoint_zeroed.pth.gz")).read()
    return torch.load(io.BytesIO(raw), map_location="cpu", weights_only=False), raw


def test_reference_pickled_modules_reproduce_golden_outputs_on_hip(golden):
    import dvg_amd.models.dcgan_64 as ours
    import dvg_amd.models.lstm as ours_lstm
    ck, _ = _load()
    enc, dec, fp = ck["encoder"], ck["decoder"], ck["frame_predictor"]
    assert isinstance(enc, ours.encoder) and isinstance(dec, ours.decoder) and isinstance(fp, ours_lstm.lstm)
    assert fp.batch_size == 50 and len(fp.hidden) == 2 and tuple(fp.hidden[0][0].shape) == (50, 256)   # pickled attributes
    # the golden case dcgan_64/eval: seeds 120 / 121 / 122 / 123 (tests/common.py)
    enc.load_state_dict(params.fill_state_dict(enc.state_dict(), 120))
    dec.load_state_dict(params.fill_state_dict(dec.state_dict(), 121, params.decoder_transposed_keys(dec.state_dict(), "dcgan")))
    enc.to(DEV).eval(), dec.to(DEV).eval()
    x = params.frames(122, 2, 1, 64).to(DEV)
    vec = params.normal(123, 2, 90, scale=0.5).tanh().to(DEV)
    with torch.no_grad():
        h, skips = enc(x)
        y, y_h = dec([vec, skips]), dec([h, skips])
    assert rel_err(h, torch.from_numpy(golden["dcgan_64/eval/h"])) < 1e-4
    assert rel_err(y, torch.from_numpy(golden["dcgan_64/eval/y"])) < 1e-4
    assert rel_err(y_h, torch.from_numpy(golden["dcgan_64/eval/y_h"])) < 1e-4
    # frame_predictor pickle -> golden lstm case (seed 300, B=5, 3 steps)
    fp.load_state_dict(params.fill_state_dict(fp.state_dict(), 300))
    fp.to(DEV).eval()
    fp.batch_size = 5
    fp.hidden = fp.init_hidden()
    with torch.no_grad():
        ys = [fp(params.normal(310 + t, 5, 90, scale=0.5).to(DEV)) for t in range(3)]
    assert rel_err(torch.stack(ys), torch.from_numpy(golden["lstm/y"])) < 1e-4
    assert rel_err(fp.hidden[1][0], torch.from_numpy(golden["lstm/h1"])) < 1e-4


def test_generate_frames_runs_from_a_reference_format_checkpoint(tmp_path):
    """generate_frames.py:43-72 on the fixture: `<model_dir>/<dataset>.pth`, opt taken from the checkpoint, GP / likelihood
    rebuilt and load_state_dict'ed from the gpytorch-style key names."""
    import generate_frames
    ck, raw = _load()
    assert sorted(ck) == ["decoder", "encoder", "frame_predictor", "gp_layer", "gp_layer_optimizer", "likelihood", "opt"]
    # give the zeroed fixture usable weights (an all-zero network is degenerate), keep the reference-format container
    ck["encoder"].load_state_dict(params.fill_state_dict(ck["encoder"].state_dict(), 120))
    ck["decoder"].load_state_dict(params.fill_state_dict(ck["decoder"].state_dict(), 121,
                                                         params.decoder_transposed_keys(ck["decoder"].state_dict(), "dcgan")))
    ck["frame_predictor"].load_state_dict(params.fill_state_dict(ck["frame_predictor"].state_dict(), 300))
    gsd, lik = params.gp_state(710)
    assert list(gsd.keys()) == list(ck["gp_layer"].keys()) and list(lik.keys()) == list(ck["likelihood"].keys())
    ck["gp_layer"], ck["likelihood"] = gsd, lik
    torch.save(ck, os.path.join(str(tmp_path), "kth.pth"))        # generate_frames.py:43 reads '<model_dir>/<dataset>.pth'
    with pytest.raises(SystemExit):                                # --data_root cannot be honoured: explicit flag required
        generate_frames.main(["--model_dir", str(tmp_path), "--dataset", "kth", "--batch_size", "4", "--nbatches", "1"])
    generate_frames.main(["--model_dir", str(tmp_path), "--dataset", "kth", "--synthetic_data", "--batch_size", "4",
                          "--n_eval", "18", "--n_future", "13", "--nsample", "2", "--nbatches", "1",
                          "--log_dir", str(tmp_path) + "/logs"])
    res = torch.load(os.path.join(str(tmp_path), "logs", "gen", "sample_lstm_0.pt"))
    assert res["psnr"].shape == (4, 2, 13) and bool(torch.isfinite(res["psnr"]).all())   # n_past = 5 from the checkpoint's opt
    assert res["best"].shape == (4,) and np.isfinite(res["ssim"].numpy()).all()
