"""GPU parity tests, module level: the HIP path of whole modules against the oracle and against the golden vectors produced by the
REFERENCE's own modules; the transparent restructurings of the rollout (hoisted skip halves, folded LSTM cell, decoder stem,
rollouts in flight, skip tensors not stored).  Split out of tests/test_gpu_parity.py in r06 (no file above 800 lines)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dvg_oracle as orc
from oracle import params
from tests.common import BACKBONE_CASES, backbone_case, dev, nhwc, oracle_backbone, rel_err, summarize, to64, yardstick

pytestmark = pytest.mark.gpu
REL = 1e-4


TRAIN_BN_BAR = 1e-4      # train-mode module outputs against the fp32 references (see the yardstick in the test)


@pytest.mark.parametrize("tag", list(BACKBONE_CASES))
def test_backbone_modules(tag, golden):
    enc, dec, esd, dsd, x, vec = backbone_case(tag)
    training = BACKBONE_CASES[tag][4]
    with torch.no_grad():
        h_ref, skips_ref, y_ref, y_h_ref, esd2, dsd2 = oracle_backbone(tag, esd, dsd, x, vec)
        enc.to(dev()), dec.to(dev())
        h, skips = enc(x.to(dev()))
        y = dec([vec.to(dev()), skips])
        y_h = dec([h, skips])
    assert h.shape == h_ref.shape and y.shape == y_ref.shape
    tol = REL
    if training:
        # Train-mode BatchNorm divides by batch statistics of B = 2-4 images: fp32 rounding of the convolutions is amplified.
        # By how much is MEASURED, not assumed: the oracle's arithmetic in fp64 is the truth, its fp32 run (and the
        # reference's own fp32 outputs, the golden) show what fp32 costs; the HIP result may be no further from the truth
        # than 4 x the fp32 oracle is (+ 3e-6).  Measured: dcgan_64 0.9-1.9 x on every output; vgg_64 2.85 x on the latent (5.6e-5
        # against the fp32 oracle's 2.0e-5; r06, native f32-MFMA build with the 256-workgroup tile thresholds: 7.5e-5 = 3.8 x).
        # r06 attributed the excess (docs/DESIGN_NOTES_r06.md section 4): not the Winograd transforms - every layer in direct form
        # is noisier - but the summation order of a K-long dot product on the matrix pipe (one accumulator per output) against
        # the host library's blocked sums, which the head's BatchNorm over four samples amplifies; it moves with the tile a
        # launch takes.  The bar against the fp32 references stays the 1e-4 of north_star (largest HIP deviation measured: 7.5e-5).
        with torch.no_grad():
            h64, skips64, y64, y_h64, _, _ = oracle_backbone(tag, to64(esd), to64(dsd), x.double(), vec.double())
        for nm, a, r32, r64 in [("h", h, h_ref, h64), ("y", y, y_ref, y64), ("y_h", y_h, y_h_ref, y_h64)] + \
                [(f"skip{i}", s_, sr, s6) for i, (s_, sr, s6) in enumerate(zip(skips, skips_ref, skips64))]:
            yardstick(f"{tag}/{nm}", a, r32, r64, ratio=4.0, slack=3e-6)
        yardstick(f"{tag}/y reference golden", y, torch.from_numpy(golden[f"{tag}/y"]), y64, ratio=4.0, slack=3e-6)
        tol = TRAIN_BN_BAR
    assert rel_err(h, h_ref) < tol, rel_err(h, h_ref)
    for s, sr in zip(skips, skips_ref):
        assert s.shape == sr.shape and rel_err(s, sr) < tol
    assert rel_err(y, y_ref) < tol and rel_err(y_h, y_h_ref) < tol
    # against the reference's own outputs
    assert rel_err(h, torch.from_numpy(golden[f"{tag}/h"])) < tol
    assert rel_err(y, torch.from_numpy(golden[f"{tag}/y"])) < tol
    assert rel_err(y_h, torch.from_numpy(golden[f"{tag}/y_h"])) < tol
    for i, s in enumerate(skips):
        np.testing.assert_allclose(summarize(s)[3:], golden[f"{tag}/skip{i}"][3:], rtol=0,
                                   atol=tol * float(np.abs(golden[f"{tag}/skip{i}"][3:]).max()) + 1e-6)
    if training:
        sd_e, sd_d = enc.state_dict(), dec.state_dict()
        for k in golden.files:
            if k.startswith(f"{tag}/enc/"):
                assert rel_err(sd_e[k.split("/enc/")[1]], torch.from_numpy(golden[k])) < 1e-4, k
            if k.startswith(f"{tag}/dec/"):
                assert rel_err(sd_d[k.split("/dec/")[1]], torch.from_numpy(golden[k])) < 1e-4, k


def test_lstm_module(golden):
    import dvg_amd.models.lstm as ours
    B = 5
    net = ours.lstm(90, 90, 256, 2, B)
    net.load_state_dict(params.fill_state_dict(net.state_dict(), 300))
    net.to(dev())
    net.hidden = net.init_hidden()
    with torch.no_grad():
        ys = [net(params.normal(310 + t, B, 90, scale=0.5).to(dev())) for t in range(3)]
    assert rel_err(torch.stack(ys), torch.from_numpy(golden["lstm/y"])) < 1e-5
    assert rel_err(net.hidden[1][0], torch.from_numpy(golden["lstm/h1"])) < 1e-5
    assert rel_err(net.hidden[1][1], torch.from_numpy(golden["lstm/c1"])) < 1e-5


def test_gaussian_lstm_module(golden):
    import dvg_amd.models.lstm as ours
    B = 5
    net = ours.gaussian_lstm(90, 90, 256, 2, B)
    net.load_state_dict(params.fill_state_dict(net.state_dict(), 310))
    net.to(dev())
    net.hidden = net.init_hidden()
    g = golden["gaussian_lstm/y"]
    with torch.no_grad():
        for t in range(3):
            z, mu, logvar = net(params.normal(320 + t, B, 90, scale=0.5).to(dev()))
            assert rel_err(mu, torch.from_numpy(g[t, 1])) < 1e-5 and rel_err(logvar, torch.from_numpy(g[t, 2])) < 1e-5
            assert z.shape == mu.shape and bool(torch.isfinite(z).all())


def test_skip_tensors_are_not_recycled():
    """SURVEY §8(b) ownership: skips returned by the encoder stay valid across later calls."""
    enc, dec, esd, dsd, x, vec = backbone_case("dcgan_64/eval")
    enc.to(dev())
    with torch.no_grad():
        h1, s1 = enc(x.to(dev()))
        keep = [s.clone() for s in s1]
        for _ in range(3):
            enc(torch.rand_like(x).to(dev()))
    assert all(torch.equal(a, b) for a, b in zip(s1, keep))


@pytest.mark.parametrize("family", ["vgg", "dcgan"])
def test_decoder_skip_hoisting_is_transparent(family):
    """Calling the eval-mode decoder repeatedly with the SAME skip tensors (a rollout) engages the hoisted skip
    halves from the second call on; every call still matches the oracle, a modified skip is recomputed, and
    DVG_SKIP_HOIST semantics (fused.SKIP_HOIST = False) give the same frames."""
    import importlib
    from dvg_amd import fused
    mod = importlib.import_module(f"dvg_amd.models.{family}_64")
    torch.manual_seed(0)
    enc, dec = mod.encoder(90, 1).to(dev()).eval(), mod.decoder(90, 1).to(dev()).eval()
    x = params.frames(110, 4, 1, 64).to(dev())
    with torch.no_grad():
        h, skip = enc(x)
        fused.clear_skip_hoist_cache()
        fused.SKIP_HOIST = False
        ref = [dec([h * s, skip]).clone() for s in (1.0, 0.5, -0.25, 0.75)]
        fused.SKIP_HOIST = True
        got = [dec([h * s, skip]).clone() for s in (1.0, 0.5, -0.25, 0.75)]
        engaged = [e for e in fused._skip_seen.values() if e[4] is not None]
        nblocks = 4 if family == "vgg" else 3   # dcgan's 4th concat layer is the last one (projection cache, ops.py)
        assert len(engaged) == nblocks and all(e[3] == 4 for e in engaged), "concat blocks hoist from the 2nd call"
        for a, b in zip(ref, got):
            assert rel_err(b, a) < 1e-5
        skip[0].mul_(0.5)                                    # in-place change of one skip tensor
        fused.SKIP_HOIST = False
        ref2 = dec([h, skip]).clone()
        fused.SKIP_HOIST = True
        assert rel_err(dec([h, skip]), ref2) < 1e-5          # first sighting of the new version: ordinary path
        assert rel_err(dec([h, skip]), ref2) < 1e-5          # second: recomputed S


def test_lstm_folded_first_cell_and_state_only_step():
    """Inference runs the first LSTMCell with the embedding folded in (dvg_lstm_cell_x: W_x = W_ih W_e); it must agree with
    the unfolded path (embed GEMM + dvg_lstm_cell, what autograd mode runs) and with the oracle, for several batch sizes
    incl. one that is not a multiple of the 8-row wave block; step_state_only() advances the state exactly like forward()."""
    import dvg_amd.models.lstm as ours
    for B in (5, 64):
        net = ours.lstm(90, 90, 256, 2, B)
        sd = params.fill_state_dict(net.state_dict(), 300)
        net.load_state_dict(sd)
        net.to(dev()).eval()
        xs = [params.normal(330 + t, B, 90, scale=0.5) for t in range(3)]
        hidden = orc.lstm_init_hidden(B, 256, 2)
        ref = [orc.lstm_step(x, sd, hidden) for x in xs]
        net.hidden = net.init_hidden()
        with torch.no_grad():
            folded = [net(x.to(dev())) for x in xs]
        h_folded = [t.clone() for pair in net.hidden for t in pair]
        net.hidden = net.init_hidden()
        unfolded = [net(x.to(dev()).requires_grad_(True)) for x in xs]     # autograd mode: embed GEMM + dvg_lstm_cell
        for a, b, r in zip(folded, unfolded, ref):
            assert rel_err(a, r) < 1e-5 and rel_err(b, r) < 1e-5 and rel_err(a, b) < 1e-5
        net.hidden = net.init_hidden()
        with torch.no_grad():
            for x in xs:
                net.step_state_only(x.to(dev()))
        for a, b in zip(h_folded, [t for pair in net.hidden for t in pair]):
            assert torch.equal(a, b)


@pytest.mark.parametrize("family", ["vgg", "dcgan"])
def test_decoder_stem_kernel_matches_generic_gemm(family):
    """Eval-mode decoder stem through dvg_stem_gemm (transposed, zero-padded weight) against the generic small-M GEMM."""
    import importlib
    from dvg_amd import fused, ops
    mod = importlib.import_module(f"dvg_amd.models.{family}_64")
    dec = mod.decoder(90, 1)
    dec.load_state_dict(params.fill_state_dict(dec.state_dict(), 77, params.decoder_transposed_keys(dec.state_dict(), family)))
    dec.to(dev()).eval()
    conv, bn = dec.upc1[0], dec.upc1[1]
    for B in (3, 64, 100):
        vec = params.normal(78, B, 90, scale=0.5).to(dev())
        with torch.no_grad():
            got = fused.stem_bn_act(conv, bn, vec)
            sc, sh = fused.folded_affine(conv, bn)
            ref = ops.gemm_nt(vec, fused.gemm_weight(conv, "stem"), sc, sh, act=ops.ACT_LRELU, slope=0.2, period=512)
        assert got.shape == (B, 512, 4, 4)
        assert rel_err(got.permute(0, 2, 3, 1).reshape(B, -1), ref) < 1e-5


def test_rollout_precomputes_frozen_skip_halves_on_a_second_stream():
    """rollout.condition() computes the decoder's loop-invariant skip halves on a side stream while the LSTM warm-up runs;
    the rollout must equal the one without hoisting, eager and as a hipGraph."""
    from dvg_amd import fused
    from dvg_amd.rollout import GraphedRollout, sample_rollout
    from tests.test_gpu_configs import _build
    B, n_past, n_eval = 8, 4, 9
    for family in ("dcgan", "vgg"):
        mods, _ = _build(family, 64, 1, B, 1900)
        for m in mods:
            m.to(dev()).eval()
        xs = [params.frames(1910 + t, B, 1, 64).to(dev()) for t in range(n_eval)]
        fused.SKIP_HOIST = False
        try:
            plain = sample_rollout(*mods, xs, n_past, n_eval, period=0)
        finally:
            fused.SKIP_HOIST = True
        fused.clear_skip_hoist_cache()
        hoisted = sample_rollout(*mods, xs, n_past, n_eval, period=0)
        assert any(e[4] is not None for e in fused._skip_seen.values()), "skip halves must have been precomputed"
        g = GraphedRollout(*mods, xs, n_past, n_eval, period=0)
        replay = [f.clone() for f in g()]
        for t in range(n_eval):
            assert rel_err(hoisted[t], plain[t]) < 2e-5 and rel_err(replay[t], plain[t]) < 2e-5, (family, t)


def test_concurrent_rollouts_equal_the_serial_chain():
    """rollout.ConcurrentRollouts: three complete rollouts in flight (one hipGraph + one stream each) must each reproduce the
    eager rollout bit for bit - no buffer may be shared between two graphs - also when replays of different graphs overlap
    many times over and new inputs are handed in between runs."""
    from dvg_amd import ops
    from dvg_amd.rollout import ConcurrentRollouts, sample_rollout
    from tests.test_gpu_configs import _build
    B, n_past, n_eval = 8, 4, 9
    for family in ("dcgan", "vgg"):
        mods, _ = _build(family, 64, 1, B, 2900)
        for m in mods:
            m.to(dev()).eval()
        xs = [params.frames(2910 + t, B, 1, 64).to(dev()) for t in range(n_eval)]
        xs2 = [params.frames(2950 + t, B, 1, 64).to(dev()) for t in range(n_eval)]
        lat = sample_rollout(*mods, xs, n_past, n_eval, period=0)
        with ops.tile_policy(True):      # chains in flight are captured with the energy-lean tiles: bit-equal under one policy
            ref = sample_rollout(*mods, xs, n_past, n_eval, period=0)
            ref2 = sample_rollout(*mods, xs2, n_past, n_eval, period=0)
        # ... and the two policies differ only by the order of the fp32 sums inside a tile
        assert max(rel_err(a, b) for a, b in zip(ref, lat)) < 5e-6
        cr = ConcurrentRollouts(*mods, xs, n_past, n_eval, inflight=3, period=0)
        outs = cr.run(11)
        torch.cuda.synchronize()
        assert len(outs) == 3
        for frames in outs:
            for t in range(n_eval):
                assert torch.equal(frames[t], ref[t]), (family, t)
        outs = cr.run(7, xs2)
        torch.cuda.synchronize()
        for frames in outs:
            for t in range(n_eval):
                assert torch.equal(frames[t], ref2[t]), (family, t)
        one = cr.run(2, xs, chains=1)
        torch.cuda.synchronize()
        assert len(one) == 1 and all(torch.equal(one[0][t], ref[t]) for t in range(n_eval))
    # with the GP trigger on (period 3: steps 6 of 4..8): every chain draws its OWN base sample per replay from the
    # captured Philox stream - frames before the trigger step equal the deterministic rollout, frames from it on differ
    # between chains and between replays of one chain
    mods, _ = _build("dcgan", 64, 1, B, 2900)
    for m in mods:
        m.to(dev()).eval()
    xs = [params.frames(2910 + t, B, 1, 64).to(dev()) for t in range(n_eval)]
    with ops.tile_policy(True):
        ref = sample_rollout(*mods, xs, n_past, n_eval, period=0)
    cr = ConcurrentRollouts(*mods, xs, n_past, n_eval, inflight=3, period=3)
    a = [[f.clone() for f in fr] for fr in cr.run(3)]
    b = [[f.clone() for f in fr] for fr in cr.run(3)]
    torch.cuda.synchronize()
    for fr in a + b:
        for t in range(7):              # frame 6 is the first one decoded from a GP sample (step i = 6)
            assert torch.equal(fr[t], ref[t]) == (t < 6), t
        assert all(bool(torch.isfinite(f).all()) for f in fr)
    assert not torch.equal(a[0][6], a[1][6]) and not torch.equal(a[1][6], a[2][6]) and not torch.equal(a[0][6], b[0][6])


def test_skip_tensors_of_part_of_a_batch_are_not_stored():
    """ABI 8 `y_from` / encoder.encode(x, skips_from=k): a rollout reads the skip tensors of ONE conditioning frame
    (generate_frames.py:154-157), so the kernels that write an encoder stage's full-resolution output beside its pooled map
    store it for the images [k, N) only.  Everything that IS returned - latent, the skips of the images [k, N), and through them
    the decoder's frames - is bit-identical to the call that stores everything, for k = 0 (all), a middle k and k = N (none);
    memory in front of / behind the shortened skip buffers is untouched (canary)."""
    from dvg_amd import fused, ops
    enc, dec, esd, dsd, _, _ = backbone_case("vgg_64/eval")
    enc.to(dev()).eval(), dec.to(dev()).eval()
    N = 32
    x = params.frames(2700, N, 1, 64).to(dev())
    with torch.no_grad():
        h_all, skips_all = enc(x)
        for k in (0, 8, 24, N):
            timer = ops.KernelTimer()
            ops.set_timer(timer)
            try:
                h, skips = enc.encode(x, skips_from=k)
            finally:
                ops.set_timer(None)
            assert torch.equal(h, h_all)
            for a, b in zip(skips, skips_all):
                if k == N:
                    assert a is None
                else:
                    assert a.shape[0] == N - k and ops.is_nhwc(a) and torch.equal(a, b[k:])
            if fused.WINOGRAD == 4 and fused.FIRST_PAIR and fused._CHAIN_LEVEL >= 2:
                # the kernels really skipped the stores: algorithmic bytes of the launches shrink by the elided images
                by = sum(v["bytes"] for v in timer.summary().values())
                if k == 0:
                    by0 = by
                else:
                    assert by < by0 - 0.9 * 4 * k * sum(s_.numel() // N for s_ in skips_all), (k, by, by0)
        # op level with canaries around the shortened buffer (a store with the wrong image offset would hit them)
        m = params.normal(2701, 36, N * 16, 128).to(dev())
        sc, sh = (1 + 0.1 * params.normal(2702, 128)).to(dev()), (0.1 * params.normal(2703, 128)).to(dev())
        from dvg_amd._lib import check, lib
        full = ops.nhwc_empty(N, 128, 16, 16, dev())
        v_full = torch.empty(36, N * 4, 128, device=dev())
        check(lib().dvg_winograd_output_pool_input(m.data_ptr(), sc.data_ptr(), sh.data_ptr(), full.data_ptr(), v_full.data_ptr(),
                                                   N, 16, 16, 128, 1, 0.2, 0, torch.cuda.current_stream().cuda_stream), "full")
        k = 20
        per = 16 * 16 * 128
        buf = torch.full(((N - k + 2) * per,), 7.25, device=dev())
        v_part = torch.empty_like(v_full)
        check(lib().dvg_winograd_output_pool_input(m.data_ptr(), sc.data_ptr(), sh.data_ptr(), buf.data_ptr() + 4 * per,
                                                   v_part.data_ptr(), N, 16, 16, 128, 1, 0.2, k,
                                                   torch.cuda.current_stream().cuda_stream), "part")
        assert torch.equal(v_part, v_full)
        assert bool((buf[:per] == 7.25).all()) and bool((buf[-per:] == 7.25).all())
        assert torch.equal(buf[per:-per], full.permute(0, 2, 3, 1).reshape(-1)[k * per:])
