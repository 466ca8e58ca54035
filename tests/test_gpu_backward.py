"""GPU parity of the TRAINING path: gradients produced by the HIP backward kernels (through
torch.autograd.Function glue) against torch autograd run on the CPU oracle for the same seeded
parameters and inputs.  Tolerance is relative to the largest gradient entry of each tensor."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dvg_oracle as orc
from oracle import params
from tests.common import BACKBONE_CASES, backbone_case, our_module, rel_err, yardstick

pytestmark = pytest.mark.gpu
GTOL = 2e-3   # batch-norm backward subtracts large nearly-equal sums: looser than the forward bar


def dev():
    return torch.device("cuda:0")


def grads_close(ours: dict, ref: dict, tol=GTOL):
    bad = []
    for k, g in ref.items():
        if g is None:
            continue
        e = rel_err(ours[k], g)
        if not e < tol:
            bad.append((k, e))
    assert not bad, bad


@pytest.mark.parametrize("N,H,C1,C2,Cout,up,pool", [
    (2, 16, 64, 0, 64, False, False), (2, 16, 64, 0, 128, False, True), (2, 16, 64, 64, 64, True, False),
    (3, 8, 128, 64, 64, True, False), (2, 32, 64, 0, 64, False, True)])
def test_conv3_block_backward(N, H, C1, C2, Cout, up, pool):
    import torch.nn as nn
    from dvg_amd import fused, ops
    hx = H // 2 if up else H
    x = params.normal(1, N, C1, hx, hx)
    sk = params.normal(2, N, C2, H, H) if C2 else None
    conv = nn.Conv2d(C1 + C2, Cout, 3, 1, 1)
    bn = nn.BatchNorm2d(Cout)
    with torch.no_grad():
        conv.weight.copy_(params.normal(3, Cout, C1 + C2, 3, 3, scale=0.05))
        conv.bias.copy_(params.normal(4, Cout, scale=0.1))
        bn.weight.copy_(1 + 0.1 * params.normal(5, Cout))
        bn.bias.copy_(0.1 * params.normal(6, Cout))
    gy = params.normal(7, N, Cout, H, H)
    gyp = params.normal(8, N, Cout, H // 2, H // 2)
    # reference: plain torch on CPU
    xr = x.clone().requires_grad_(True)
    skr = sk.clone().requires_grad_(True) if sk is not None else None
    ps = {k: v.clone().requires_grad_(True) for k, v in (("w", conv.weight.data), ("b", conv.bias.data),
                                                          ("g", bn.weight.data), ("be", bn.bias.data))}
    xin = F.interpolate(xr, scale_factor=2, mode="nearest") if up else xr
    if skr is not None:
        xin = torch.cat([xin, skr], 1)
    yr = F.leaky_relu(F.batch_norm(F.conv2d(xin, ps["w"], ps["b"], 1, 1), None, None, ps["g"], ps["be"], True, 0.1,
                                   1e-5), 0.2)
    loss = (yr * gy).sum()
    if pool:
        loss = loss + (F.max_pool2d(yr, 2, 2) * gyp).sum()
    loss.backward()
    # ours
    conv.to(dev()), bn.to(dev())
    xo = ops.to_nhwc(x.to(dev())).requires_grad_(True)
    sko = ops.to_nhwc(sk.to(dev())).requires_grad_(True) if sk is not None else None
    out = fused.conv3_bn_act(conv, bn, xo, sko, upsample=up, pool=pool)
    if pool:
        lo = (out[0] * gy.to(dev())).sum() + (out[1] * gyp.to(dev())).sum()
    else:
        lo = (out * gy.to(dev())).sum()
    lo.backward()
    assert rel_err(out[0] if pool else out, yr) < 1e-4
    ours = {"x": xo.grad, "w": conv.weight.grad, "b": conv.bias.grad, "g": bn.weight.grad, "be": bn.bias.grad}
    ref = {"x": xr.grad, "w": ps["w"].grad, "g": ps["g"].grad, "be": ps["be"].grad}
    if sk is not None:
        ours["sk"], ref["sk"] = sko.grad, skr.grad
    grads_close(ours, ref)
    assert float(conv.bias.grad.abs().max()) < 1e-3 * float(ps["w"].grad.abs().max()) + 1e-6  # ~0 under batch stats


@pytest.mark.parametrize("N,H,Cin,Cout,items", [(8, 8, 128, 128, 1), (3, 8, 128, 256, 2), (4, 16, 256, 128, 3),
                                                 (2, 32, 128, 128, 1), (8, 8, 512, 512, 8)])
def test_winograd_weight_gradient_matches_direct_and_fp64(N, H, Cin, Cout, items):
    """dvg_winograd_wgrad_* (F(4x4,3x3) form, several uses of a layer in one product) against the direct kernel's slabs and
    against the fp64 weight gradient of F.conv2d; the (3, 8, ...) case has 24 tiles: the zero-padded K tail."""
    from dvg_amd import ops
    xs = [ops.to_nhwc(params.normal(20 + i, N, Cin, H, H).to(dev())) for i in range(items)]
    dus = [ops.to_nhwc(params.normal(40 + i, N, Cout, H, H).to(dev())) for i in range(items)]
    assert ops.winograd_wgrad_ok(N, Cin, H, H, Cout)
    packed = ops.winograd_wgrad_partial_multi(xs, dus)
    assert tuple(packed.shape) == (1, 9, Cout, Cin)
    direct = ops.conv_wgrad_partial_multi(ops.MODE_CONV3, xs, None, dus).sum(0, keepdim=True)
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    sum((F.conv2d(x.double().cpu(), w, None, 1, 1) * du.double().cpu()).sum() for x, du in zip(xs, dus)).backward()
    ref = w.grad.permute(2, 3, 0, 1).reshape(1, 9, Cout, Cin)
    assert rel_err(direct, ref) < 2e-5
    assert rel_err(packed, ref) < 2e-4            # F(4x4,3x3)'s transform constants cost ~1 digit in fp32
    # through the in-place finish: beta accumulates
    dst = torch.ones(Cout, Cin, 3, 3, device=dev())
    ops.wgrad_finish(packed, dst, 0, 3, 3, beta=1.0)
    assert rel_err(dst - 1.0, w.grad) < 2e-4


@pytest.mark.parametrize("N,H,Cin,Cout,items", [(32, 8, 128, 128, 3), (8, 16, 256, 128, 2)])
def test_winograd_weight_gradient_from_saved_input_transforms(N, H, Cin, Cout, items):
    """The forward's input transforms (conv3x3_winograd(return_v=True)) read in place, one buffer per use, give the same slab
    as recomputing them from the layer inputs."""
    from dvg_amd import ops
    xs = [ops.to_nhwc(params.normal(60 + i, N, Cin, H, H).to(dev())) for i in range(items)]
    dus = [ops.to_nhwc(params.normal(70 + i, N, Cout, H, H).to(dev())) for i in range(items)]
    u = ops.winograd_weight(params.normal(80, Cout, Cin, 3, 3, scale=0.05).to(dev()), 4)
    vs = [ops.conv3x3_winograd(x, u, None, None, act=ops.ACT_NONE, return_v=True)[1] for x in xs]
    assert tuple(vs[0].shape) == (36, N * (H // 4) ** 2, Cin)
    a = ops.winograd_wgrad_partial_multi(xs, dus)
    b = ops.winograd_wgrad_partial_multi(xs, dus, vs)
    assert torch.equal(a, b)        # same operands, same kernel, same order of operations
    # a V that does not fit (wrong shape) falls back to recomputation
    c = ops.winograd_wgrad_partial_multi(xs, dus, [v[:, :64] for v in vs])
    assert torch.equal(a, c)


@pytest.mark.parametrize("rows,C", [(64, 90), (1216, 1024), (7, 33), (100, 16)])
def test_colsum_matches_torch(rows, C):
    """dvg_colsum (bias gradients; rows = batch x time steps when a BPTT pass is reduced in one launch), plain and accumulating."""
    from dvg_amd import ops
    a = params.normal(90, rows, C).to(dev())
    ref = a.double().sum(0)
    out = ops.colsum(a)
    assert float((out.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
    acc = torch.full((C,), 2.0, device=dev())
    ops.colsum(a, out=acc, accumulate=True)
    assert float((acc.double() - 2.0 - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))


def _reference_case(family, seed):
    """The B=16 train-mode case of tests/golden/make_golden.py:run_backbone_grads, rebuilt from its seeds."""
    mod = our_module(family, 64)
    enc, dec = mod.encoder(90, 1), mod.decoder(90, 1)
    esd = params.fill_state_dict(enc.state_dict(), seed)
    dsd = params.fill_state_dict(dec.state_dict(), seed + 1, params.decoder_transposed_keys(dec.state_dict(), family))
    enc.load_state_dict(esd), dec.load_state_dict(dsd)
    x = params.frames(seed + 2, 16, 1, 64)
    gy, gh = params.normal(seed + 4, 16, 1, 64, 64), params.normal(seed + 5, 16, 90)
    return enc, dec, esd, dsd, x, gy, gh


def _oracle_grads(family, esd, dsd, x, gy, gh, dtype):
    def mk(sd):
        out = {}
        for k, v in sd.items():
            if v.is_floating_point():
                v = v.to(dtype).clone()
                if "running" not in k:
                    v.requires_grad_(True)
            out[k] = v
        return out
    e, d = mk(esd), mk(dsd)
    if family == "vgg":
        h, skips = orc.vgg_encoder(x.to(dtype), e, True)
        y = orc.vgg_decoder(h, skips, d, True)
    else:
        h, skips = orc.dcgan_encoder(x.to(dtype), e, True)
        y = orc.dcgan_decoder(h, skips, d, True, "tanh")
    ((y * gy.to(dtype)).sum() + (h * gh.to(dtype)).sum()).backward()
    return h.detach(), y.detach(), e, d


def _hip_grads(family, seed, record=False):
    """encoder -> decoder([h, skips]) in train mode at B=16 on the HIP path, `.backward()` of sum(y*Gy) + sum(h*Gh)."""
    from tests.common import record_hip_kinks, single_call_kinks
    enc, dec, esd, dsd, x, gy, gh = _reference_case(family, seed)
    enc.to(dev()).train(), dec.to(dev()).train()
    with record_hip_kinks(enc, dec) as acts:
        ho, so = enc(x.to(dev()))
        yo = dec([ho, so])
    ((yo * gy.to(dev())).sum() + (ho * gh.to(dev())).sum()).backward()
    grads = {"enc": {k: p.grad.double().cpu() for k, p in enc.named_parameters()},
             "dec": {k: p.grad.double().cpu() for k, p in dec.named_parameters()}}
    return (esd, dsd, x, gy, gh), ho.detach().cpu(), yo.detach().cpu(), grads, single_call_kinks(acts)


def _tensor_errors(ours, ref):
    """(max |diff| / max |ref|, ||diff|| / ||ref||) over ALL entries of a gradient tensor."""
    diff = ours.double() - ref.double()
    return (float(diff.abs().max()) / max(float(ref.abs().max()), 1e-30),
            float(diff.norm() / ref.double().norm().clamp_min(1e-30)))


KINK_MAX, KINK_L2 = 2e-4, 1e-4      # measured r04: dcgan_64 2.7e-6 / 4.4e-6, vgg_64 3.1e-5 / 4.5e-5 (VERDICT r03 asked for 1e-3 / 2e-4)


@pytest.mark.parametrize("family,seed", [("dcgan", 200), ("vgg", 210)])
def test_module_backward_matches_reference_gradients(family, seed, golden):
    """encoder -> decoder([h, skips]) in train mode at B=16, `.backward()` of sum(y*Gy) + sum(h*Gh): EVERY parameter gradient,
    EVERY entry (nothing dropped), against the fp64 autograd of the oracle - which tests/test_oracle_golden.py pins on CPU to
    the gradients of the reference's own modules and `.backward()` - at max error <= 2e-4 of the tensor's largest entry and
    L2 error <= 1e-4 of its norm.

    The oracle takes its LeakyReLU / max-pool BRANCHES from the HIP forward (oracle.forced_kinks fed by
    tests.common.record_hip_kinks: slope from the sign of the layer output the kernels saved, pool winner = its arg-max);
    all arithmetic stays the oracle's.  A pre-activation within rounding of 0 (or a pool window whose two largest entries are
    within rounding of each other) otherwise sends two implementations that agree to 1e-7 in the FORWARD down different
    branches, and each such flip moves individual gradient entries by percents (22 layers of batch-statistics BatchNorm:
    the reference's own arithmetic in fp32 deviates from its fp64 run by 4.6e-2 max / 5.5e-3 L2 for that reason alone, and by
    3e-5 / 1.3e-5 once the branches are shared - tests/test_oracle_golden.py::test_forced_kinks_isolate_backward_arithmetic).
    With the branches shared a defect that hits one entry per tile is visible at the 2e-4 bar.  The free-running comparison
    (different branches, noise-limited) is the statistical test below.  Also: the forward against the reference's golden
    latent and the fp64 frame at 1e-4, and for dcgan_64 (10 layers: flips do not dominate) the reference's own fp32 gradient
    fingerprints (tests/golden) at 1e-2 / 2e-3.  Conv biases that feed a train-mode BatchNorm have an analytically zero
    gradient and are skipped."""
    from tests.test_oracle_golden import fingerprint_errors, is_bn_fed_conv_bias
    (esd, dsd, x, gy, gh), ho, yo, grads, acts = _hip_grads(family, seed)
    with orc.forced_kinks({k: v.double() for k, v in acts.items()}):
        h64, y64, e64, d64 = _oracle_grads(family, esd, dsd, x, gy, gh, torch.float64)
    tag = f"{family}_64/grad"
    # the train-mode latent (through 13 batch-statistics BatchNorm layers at B = 16) against the reference's own fp32 latent and
    # against the fp64 oracle: 1e-4 on the product build.  The native-f32-MFMA build of the library (DVG_HIP_LIB=...f32mfma.so: the
    # less exact arithmetic; a comparison build, never the measured one) sits AT that bar on vgg_64 seed 210 - 1.14e-4 /
    # 1.10e-4 after r06's weight-transform kernels changed U in the last place, below 1e-4 before - and gets 1.5e-4.  Frames: 1e-4.
    from dvg_amd._lib import lib
    h_bar = 1e-4 if lib().dvg_mfma_mode() == 1 else 1.5e-4
    assert rel_err(ho, torch.from_numpy(golden[f"{tag}/h"])) < h_bar
    assert rel_err(yo, y64) < 1e-4 and rel_err(ho, h64) < h_bar
    bad, n, worst = [], 0, [0.0, 0.0]
    for name, r64 in (("enc", e64), ("dec", d64)):
        for k, g in grads[name].items():
            if is_bn_fed_conv_bias(k):
                continue
            err, l2 = _tensor_errors(g, r64[k].grad)
            n += 1
            worst = [max(worst[0], err), max(worst[1], l2)]
            ok = err < KINK_MAX and l2 < KINK_L2
            if family == "dcgan":       # the reference's own fp32 gradients, branches free on both sides
                err_f, l2_f, sq_f = fingerprint_errors(g, golden[f"{tag}/{name}/{k}"])
                ok = ok and err_f < 2e-2 and l2_f < 4e-3 and sq_f < 8e-3
            if not ok:
                bad.append((name, k, err, l2))
    assert n >= (14 if family == "dcgan" else 40)
    print(f"{family}: forced kinks: worst max-err {worst[0]:.2e}, worst L2 {worst[1]:.2e} over {n} tensors")
    assert not bad, [tuple(f"{v:.3e}" if isinstance(v, float) else v for v in b) for b in bad[:8]]


@pytest.mark.slow
def test_vgg_backward_free_running_noise_is_the_fp32_noise(golden):
    """The second, statistical check of vgg_64's 22-layer backward: branches FREE on every side (HIP, the oracle in fp32, the
    oracle in fp64), seeds 210-214.  Per tensor the deviation from the fp64 run is then dominated by WHICH near-zero
    pre-activations flip - a property of each side's forward rounding pattern, and proportional to the size of its forward
    rounding error (the number of pre-activations that land on the other side of 0).  Over all tensors of all seeds, HIP's
    median and 95th-percentile per-tensor errors (max entry and L2) are compared with the fp32 oracle's (the reference's own
    arithmetic in fp32 on the CPU):
      * with the 3x3 layers in their direct implicit-GEMM form (fused.WINOGRAD = 0) the HIP path's forward rounding is
        0.8-2.4e-6 of a layer's largest output where torch's blocked CPU sums have 0.4-0.9e-6 (a sequential K loop of up to
        9216 fp32 accumulations; DESIGN.md 3.1d's table): measured 1.6-2.1 x the fp32 oracle's noise, bar 2.5 x;
      * in the default form the layers on maps up to 32 x 32 run as Winograd F(4x4,3x3), whose transforms round at ~1e-5 of a
        layer's largest output (tests/test_gpu_parity.py: within the 1e-4 forward bar) - proportionally more flips, measured
        2.6-3.2 x the fp32 oracle's noise: bar 4 x.
    The arithmetic of the backward itself is pinned by the forced-branch test above (3e-5 / 4.5e-5 with nothing dropped);
    this one shows that what remains free-running is branch noise of the expected size.  Also: the reference's own gradient
    fingerprints of seed 210 (tests/golden: fp32 with its own flips) in the sum of squares."""
    from dvg_amd import fused
    from tests.test_oracle_golden import fingerprint_errors, is_bn_fed_conv_bias
    hip, cpu = {4: [], 0: []}, []
    for seed in range(210, 215):
        ref = None
        for wino in (4, 0):
            old, fused.WINOGRAD = fused.WINOGRAD, min(wino, fused.WINOGRAD)
            try:
                (esd, dsd, x, gy, gh), ho, yo, grads, _ = _hip_grads("vgg", seed)
            finally:
                fused.WINOGRAD = old
            if ref is None:
                _, _, e64, d64 = _oracle_grads("vgg", esd, dsd, x, gy, gh, torch.float64)
                _, _, e32, d32 = _oracle_grads("vgg", esd, dsd, x, gy, gh, torch.float32)
                ref = True
                for name, r64, r32 in (("enc", e64, e32), ("dec", d64, d32)):
                    for k in grads[name]:
                        if not is_bn_fed_conv_bias(k):
                            cpu.append(_tensor_errors(r32[k].grad, r64[k].grad))
            for name, r64 in (("enc", e64), ("dec", d64)):
                for k, g in grads[name].items():
                    if is_bn_fed_conv_bias(k):
                        continue
                    hip[wino].append(_tensor_errors(g, r64[k].grad))
                    if seed == 210 and wino == 4:
                        sq_f = fingerprint_errors(g, golden[f"vgg_64/grad/{name}/{k}"])[2]
                        assert sq_f < 4e-2, (k, sq_f)
    cpu = np.array(cpu)
    # the native f32-MFMA build (DVG_HIP_LIB=...f32mfma.so) rounds its running sum after every 2 products, the bf16-triple form
    # after 16: its forward error is 1.3-3.2e-6 where the triples have 0.8-2.4e-6, measured 1.8-2.8 x the fp32
    # oracle's noise in the direct form - proportionally wider bars there
    from dvg_amd import _lib
    bars = ((0, 2.5), (4, 4.0)) if _lib.lib().dvg_mfma_mode() == 1 else ((0, 3.5), (4, 5.0))
    for wino, bar in bars:
        h = np.array(hip[wino])
        stats = {}
        for j, what in enumerate(("max", "l2")):
            for q in (50, 95):
                stats[(what, q)] = (float(np.percentile(h[:, j], q)), float(np.percentile(cpu[:, j], q)))
        print(f"vgg_64 free-running noise, WINOGRAD={wino} (HIP, fp32 oracle):",
              {k: (f"{a:.2e}", f"{b:.2e}", f"{a / b:.2f}x") for k, (a, b) in stats.items()})
        for k, (a, b) in stats.items():
            assert a <= bar * b, (wino, k, a, b)


LSTM_BAR = 3e-6         # LSTM BPTT gradients against the fp32 references: measured r05 HIP vs fp32 oracle <= 6.3e-7, HIP vs fp64
#                         <= 2.9e-7 (the fp32 oracle itself: <= 5.8e-7) - r04's bar was 1e-3


def test_lstm_bptt_matches_reference_gradients(golden):
    """4-step BPTT through lstm.lstm at B=16 against the reference's own backward (tests/golden lstm_grad/*)."""
    import dvg_amd.models.lstm as ours
    from tests.test_oracle_golden import fingerprint_errors
    B, seed = 16, 320
    net = ours.lstm(90, 90, 256, 2, B)
    net.load_state_dict(params.fill_state_dict(net.state_dict(), seed))
    net.to(dev())
    net.hidden = net.init_hidden()
    xs = [params.normal(seed + 10 + t, B, 90, scale=0.5).to(dev()).requires_grad_(True) for t in range(4)]
    loss = sum((net(xs[t]) * params.normal(seed + 20 + t, B, 90).to(dev())).sum() for t in range(4))
    loss.backward()
    assert abs(float(loss) - float(golden["lstm_grad/loss"][0])) < 1e-4
    # yardstick: the same BPTT by the oracle's autograd in fp64 (truth) and fp32 (what the reference's arithmetic costs)
    sd = params.fill_state_dict(ours.lstm(90, 90, 256, 2, B).state_dict(), seed)

    def oracle_bptt(dt):
        leaf = {k: v.to(dt).clone().requires_grad_(True) for k, v in sd.items()}
        hid = orc.lstm_init_hidden(B, 256, 2, dtype=dt)
        xr = [params.normal(seed + 10 + t, B, 90, scale=0.5).to(dt).requires_grad_(True) for t in range(4)]
        sum((orc.lstm_step(xr[t], leaf, hid) * params.normal(seed + 20 + t, B, 90).to(dt)).sum() for t in range(4)).backward()
        return {k: v.grad for k, v in leaf.items()}, [t.grad for t in xr]
    (g32, x32), (g64, x64) = oracle_bptt(torch.float32), oracle_bptt(torch.float64)
    for k, p in net.named_parameters():
        yardstick(f"lstm BPTT grad {k}", p.grad, g32[k], g64[k], ratio=1.5, slack=2e-7)
        assert rel_err(p.grad, g32[k]) < LSTM_BAR, (k, rel_err(p.grad, g32[k]))
        err, l2, sq = fingerprint_errors(p.grad, golden[f"lstm_grad/{k}"])          # the REFERENCE's own gradients (fp32)
        assert err < 1e-5 and l2 < 1e-5 and sq < 2e-5, (k, err, l2, sq)      # (64 strided samples + three sums per tensor)
    for t in range(4):
        yardstick(f"lstm BPTT grad x{t}", xs[t].grad, x32[t], x64[t], ratio=1.5, slack=2e-7)
        assert rel_err(xs[t].grad, torch.from_numpy(golden[f"lstm_grad/x{t}"])) < LSTM_BAR


def test_lstm_bptt_backward():
    import dvg_amd.models.lstm as ours
    B = 6
    net = ours.lstm(90, 90, 256, 2, B)
    sd = params.fill_state_dict(net.state_dict(), 300)
    net.load_state_dict(sd)
    ref = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xs = [params.normal(400 + t, B, 90, scale=0.5) for t in range(4)]
    gs = [params.normal(410 + t, B, 90) for t in range(4)]
    hidden = orc.lstm_init_hidden(B, 256, 2)
    xr = [t.clone().requires_grad_(True) for t in xs]
    loss = sum((orc.lstm_step(xr[t], ref, hidden) * gs[t]).sum() for t in range(4))
    loss.backward()
    net.to(dev())
    net.hidden = net.init_hidden()
    xo = [t.to(dev()).requires_grad_(True) for t in xs]
    lo = sum((net(xo[t]) * gs[t].to(dev())).sum() for t in range(4))
    lo.backward()
    assert abs(float(lo) - float(loss)) < 1e-5 * abs(float(loss)) + 1e-4
    grads_close({k: p.grad for k, p in net.named_parameters()}, {k: v.grad for k, v in ref.items()}, tol=1e-5)
    for t in range(4):
        assert rel_err(xo[t].grad, xr[t].grad) < 1e-5


@pytest.mark.parametrize("B,S", [(16, 4), (50, 14), (6, 3), (64, 19)])
def test_lstm_sequence_matches_the_step_by_step_module(B, S):
    """models.lstm.forward_sequence (autograd._LSTMSequence: a teacher-forced sequence with ONE GEMM per non-recurrent
    product over all S x B rows - embedding, both cells' input halves, output head, their data and weight gradients - and
    one dvg_lstm_cell_pre / dvg_lstm_cell_bwd launch per step and layer) against S calls of the module (the path that
    tests/golden pins to the reference's own BPTT): outputs and every gradient - all 12 parameter tensors and the inputs -
    under a per-step weighting, B = 50 / 6 (not multiples of 8: the clamped rows of both kernels), S up to 19 (C2)."""
    import dvg_amd.models.lstm as ours
    res = {}
    for seq in (False, True):
        net = ours.lstm(90, 90, 256, 2, B)
        net.load_state_dict(params.fill_state_dict(net.state_dict(), 330))
        net.to(dev())
        assert ours.sequence_applies(net)
        xs = torch.stack([params.normal(340 + t, B, 90, scale=0.5) for t in range(S)]).to(dev()).requires_grad_(True)
        ws = torch.stack([params.normal(370 + t, B, 90) for t in range(S)]).to(dev())
        net.hidden = net.init_hidden()
        if seq:
            y = ours.forward_sequence(net, xs)
        else:
            y = torch.stack([net(xs[t]) for t in range(S)])
        (y * ws).sum().backward()
        g = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
        g["x"] = xs.grad.detach().clone()
        res[seq] = (y.detach().clone(), g)
    (ya, ga), (yb, gb) = res[False], res[True]
    assert yb.shape == (S, B, 90) and rel_err(yb, ya) < 2e-6, rel_err(yb, ya)
    assert len(gb) == 13
    for k in ga:
        assert rel_err(gb[k], ga[k]) < 2e-5, (k, rel_err(gb[k], ga[k]))


def test_lstm_sequence_kernels_against_their_unfused_forms():
    """dvg_lstm_cell_pre == dvg_lstm_cell with the input half handed over as a pre-activation; dvg_lstm_cell_bwd ==
    dvg_lstm_gates_bwd followed by the dG W_hh GEMM, with both / one / none of (dh_b, dc), the zero initial state as a NULL
    c_prev, no dh_prev at the first step, and a batch that is not a multiple of 8."""
    from dvg_amd import ops
    B, H = 21, 256
    x, h, c = (params.normal(390 + i, B, H, scale=0.6).to(dev()) for i in range(3))
    w_ih, w_hh = (params.normal(394 + i, 4 * H, H, scale=0.06).to(dev()) for i in range(2))
    b_ih, b_hh = (params.normal(396 + i, 4 * H, scale=0.1).to(dev()) for i in range(2))
    h2, c2, gates = ops.lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh, want_gates=True)
    pre = ops.gemm_nt(x, w_ih, None, b_ih + b_hh)
    h3, c3, g3 = torch.empty_like(h), torch.empty_like(c), torch.empty_like(gates)
    ops.lstm_cell_pre(pre, h, c, w_hh, h3, c3, g3)
    assert rel_err(h3, h2) < 2e-6 and rel_err(c3, c2) < 2e-6 and rel_err(g3, gates) < 2e-6
    dh_a, dh_b, dc = (params.normal(400 + i, B, H).to(dev()) for i in range(3))
    w_hh_t = w_hh.t().contiguous()
    for use_b, use_dc, c_prev, want_dh in ((True, True, c, True), (False, True, c, True), (True, False, None, True),
                                           (False, False, c, False)):
        dh = dh_a + dh_b if use_b else dh_a
        cp = c if c_prev is not None else torch.zeros_like(c)
        _, c2z, gz = ops.lstm_cell(x, h, cp, w_ih, w_hh, b_ih, b_hh, want_gates=True)
        dG_ref, dcp_ref = ops.lstm_gates_bwd(dh, dc if use_dc else None, gz, cp, c2z)
        dhp_ref = ops.gemm_nt(dG_ref, w_hh_t, None, None)
        dG, dcp = torch.empty_like(dG_ref), torch.empty_like(dcp_ref)
        dhp = torch.empty_like(dhp_ref) if want_dh else None
        ops.lstm_cell_bwd(dh_a, dh_b if use_b else None, dc if use_dc else None, gz, c_prev, c2z, w_hh_t, dG, dcp, dhp)
        assert rel_err(dG, dG_ref) < 2e-6 and rel_err(dcp, dcp_ref) < 2e-6, (use_b, use_dc)
        if want_dh:
            assert rel_err(dhp, dhp_ref) < 5e-6, rel_err(dhp, dhp_ref)


@pytest.mark.parametrize("B,D,M", [(16, 12, 40), (64, 90, 40), (50, 7, 24), (128, 8, 40), (101, 5, 40), (72, 3, 40)])
def test_gp_train_backward(B, D, M):
    """dvg_gp_train_bwd through gp_autograd against fp64 autograd of the oracle, every gradient at 1e-4.  B = 128 / 101 / 72:
    the data points in two chunks (64 + 64, 51 + 50, 36 + 36) - since r04 no shape reachable from `train.py --batch_size`
    (<= 128) leaves the fp64 arithmetic at M = 40 (until r03: fp32 above B = 71, bars 5e-4 / 2e-3)."""
    from dvg_amd import _lib
    assert _lib.lib().dvg_gp_bwd_precision(B, M) == 64
    assert (_lib.lib().dvg_gp_bwd_chunk(B, M) < B) == (B > 71)
    from dvg_amd.models.gp_models import GaussianLikelihood, GPRegressionLayer1, VariationalELBO
    sd, lik = params.gp_state(500, D=D, M=M)
    h = params.normal(501, B, D, scale=0.7).tanh()
    tgt = params.normal(502, D, B, scale=0.5)
    gmean = params.normal(503, D, B)
    # oracle fp64 autograd
    rs = {k: v.double().clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point()}
    rl = {k: v.double().clone().requires_grad_(True) for k, v in lik.items()}
    hr = h.double().clone().requires_grad_(True)
    full = dict(sd)
    full.update(rs)
    pr = orc.gp_predict(hr, full, training=True)
    elbo = orc.variational_elbo(pr, tgt.double(), orc.likelihood_noise(rl), num_data=B)
    loss_r = -(elbo.sum()) + (pr["mean"] * gmean.double()).sum()
    loss_r.backward()
    # ours
    gp, like = GPRegressionLayer1(D, M), GaussianLikelihood(batch_size=D)
    gp.load_state_dict(sd)
    like.load_state_dict(lik)
    gp.to(dev()).train(), like.to(dev()).train()
    mll = VariationalELBO(like, gp, num_data=B)
    ho = h.to(dev()).requires_grad_(True)
    pred = gp(ho.transpose(0, 1).view(D, B, 1))
    loss_o = -(mll(pred, tgt.to(dev())).sum()) + (pred.mean * gmean.to(dev())).sum()
    loss_o.backward()
    assert abs(float(loss_o) - float(loss_r)) < 1e-5 * abs(float(loss_r)) + 1e-5
    ours = {k: p.grad for k, p in gp.named_parameters()}
    ours["h"] = ho.grad
    ours["noise"] = like.noise_covar.raw_noise.grad
    ref = {k: v.grad for k, v in rs.items()}
    ref["h"] = hr.grad
    ref["noise"] = rl["noise_covar.raw_noise"].grad
    ref["variational_strategy.variational_distribution.chol_variational_covar"] = torch.tril(
        ref["variational_strategy.variational_distribution.chol_variational_covar"])
    grads_close(ours, ref, tol=1e-4)     # fp64 inside dvg_gp_train_bwd (r02, fp32 throughout: 5e-3)


def test_fused_elbo_equals_the_torch_composition():
    """VariationalELBO.forward as dvg_gp_elbo / dvg_gp_elbo_bwd (one launch each way) against the same expression composed term
    by term in torch ops (likelihood.expected_log_prob / B - KL / num_data): values and every gradient - GP parameters and h
    through mean / variance / KL, the raw noise through the soft-plus, and the TARGET handed over as the strided view
    h_target.transpose(0, 1) (train.py:226)."""
    from dvg_amd.models import gp_models as gm
    B, D, M = 16, 90, 40
    sd, lik = params.gp_state(520, D=D, M=M)
    h = params.normal(521, B, D, scale=0.7).tanh()
    ht = params.normal(522, B, D, scale=0.5)
    w = params.normal(523, D).to(dev())
    res = {}
    for fused_on in (True, False):
        gp, like = gm.GPRegressionLayer1(D, M), gm.GaussianLikelihood(batch_size=D)
        gp.load_state_dict(sd)
        like.load_state_dict(lik)
        gp.to(dev()).train(), like.to(dev()).train()
        mll = gm.VariationalELBO(like, gp, num_data=37)
        ho = h.to(dev()).requires_grad_(True)
        hto = ht.to(dev()).requires_grad_(True)
        pred = gp(ho.transpose(0, 1).view(D, B, 1))
        if fused_on:
            elbo = mll(pred, hto.transpose(0, 1))
        else:
            elbo = like.expected_log_prob(hto.transpose(0, 1), pred) / B - pred.kl / 37
        (elbo * w).sum().backward()
        g = {k: p.grad.clone() for k, p in gp.named_parameters()}
        g.update(h=ho.grad.clone(), target=hto.grad.clone(), noise=like.noise_covar.raw_noise.grad.clone())
        res[fused_on] = (elbo.detach().clone(), g)
    (ea, ga), (eb, gb) = res[True], res[False]
    assert ea.shape == (D,) and rel_err(ea, eb) < 1e-5
    for k in ga:
        assert rel_err(ga[k], gb[k]) < 2e-4, (k, rel_err(ga[k], gb[k]))


@pytest.mark.parametrize("R,M,N", [(176, 1024, 256), (60, 90, 256), (176, 256, 90), (1, 7, 5), (33, 31, 65), (600, 64, 36)])
def test_gemm_tn_matches_fp64_reference(R, M, N):
    """ops.gemm_tn (dW = dY^T X of the dense layers in one launch, bias gradients as column sums of dY riding along) against
    fp64 matmul: fresh output, accumulation into existing buffers, row-sliced operands (the BPTT pass hands over dG[B:]),
    widths that are not multiples of 4 (the 90-wide latent code), and the long-sum fallback (R > ops.GEMM_TN_MAX_ROWS)."""
    from dvg_amd import ops
    a = params.normal(570, R + 3, M).to(dev())[3:]
    b = params.normal(571, R + 3, N).to(dev())[3:]
    ref = a.double().t() @ b.double()
    cref = a.double().sum(0)
    out = ops.gemm_tn(a, b)
    assert out.shape == (M, N) and rel_err(out, ref) < 2e-6, rel_err(out, ref)
    base, c0, c1 = params.normal(572, M, N).to(dev()), params.normal(573, M).to(dev()), params.normal(574, M).to(dev())
    o2, k0, k1 = base.clone(), c0.clone(), c1.clone()
    assert ops.gemm_tn(a, b, out=o2, accumulate=True, colsums=(k0, None, k1)) is o2
    assert rel_err(o2, base.double() + ref) < 2e-6
    assert rel_err(k0, c0.double() + cref) < 2e-6 and rel_err(k1, c1.double() + cref) < 2e-6
    k2 = torch.full((M,), 7.0, device=dev())
    ops.gemm_tn(a, b, out=o2, colsums=(k2,), colsum_accumulate=False)
    assert rel_err(o2, ref) < 2e-6 and rel_err(k2, cref) < 2e-6


@pytest.mark.parametrize("S,B,k", [(11, 16, 6), (15, 4, 8), (7, 16, 4), (3, 40, 2), (5, 8, 5), (4, 16, 3)])
def test_gp_step_groups_equal_one_workgroup_per_step(S, B, k):
    """dvg_gp_predict / dvg_gp_train_bwd with step_group = k (k time steps of a latent dim as ONE workgroup's problem of k x B
    points: K_ZZ, its factor and the KL term once per group) against one workgroup per (step, dim): mean and variance
    bit-identical (a point's arithmetic does not depend on its neighbours), KL equal after the fp32 rounding, every
    gradient at 1e-5 (the points' contributions are summed in another order).  Shapes: the C4 / C5 closures (B = 16, S = 11:
    two chunks of 48 points backward; B = 4, S = 15), ragged last groups in the one-chunk layout (7 = 4 + 3) and in the
    chunked layout with fewer points than a chunk (B = 40: 2 + 1), one group for all steps, and the heuristic's own choice."""
    from dvg_amd import ops
    from dvg_amd.models import gp_models as gm
    D, M = 90, 40
    sd, _ = params.gp_state(560, D=D, M=M)
    gp = gm.GPRegressionLayer1(D, M)
    gp.load_state_dict(sd)
    gp.to(dev())
    gp.ensure_initialized()
    vs = gp.variational_strategy
    s_, ell, c = [t.detach() for t in gp.hypers()]
    par = (vs.inducing_points.detach().squeeze(-1), vs.variational_distribution.variational_mean.detach(),
           vs.variational_distribution.chol_variational_covar.detach(), c, s_, ell)
    h = params.normal(561, B, S * D, scale=0.7).tanh().to(dev())
    gmean, gvar = params.normal(562, S * D, B).to(dev()), params.normal(563, S * D, B).abs().to(dev())
    gkl = params.normal(564, S * D).to(dev())
    assert ops.gp_step_group(16, 11, 90, 40) == 6 and ops.gp_step_group(4, 15, 90, 40) == 8      # C4, C5
    assert ops.gp_step_group(64, 19, 90, 40) == 1 and ops.gp_step_group(16, 2, 90, 40) == 1      # no fit / one round anyway
    out = {}
    for kk in (1, k):
        r = ops.gp_predict(h, *par, want_var=True, want_kl=True, train_mode=True, param_period=D, step_group=kk)
        g = ops.gp_train_bwd(h, *par, gmean, gvar, gkl, param_period=D, step_group=kk)
        assert g["groups"] == -(-S // kk) and g["dz"].shape[0] == g["groups"] * D
        names = ("dz", "dm", "dls", "dc", "ds", "dell")
        summed = ops.sum_steps([g[n] for n in names], g["groups"]) if g["groups"] > 1 else [g[n].reshape(-1) for n in names]
        out[kk] = (r, g["dh"], dict(zip(names, summed)))
    (ra, dha, ga), (rb, dhb, gb) = out[1], out[k]
    assert torch.equal(ra["mean"], rb["mean"]) and torch.equal(ra["var"], rb["var"])
    assert rel_err(rb["kl"], ra["kl"]) < 1e-6
    assert rel_err(dhb, dha) < 1e-5, rel_err(dhb, dha)
    for n in ga:
        assert rel_err(gb[n], ga[n]) < 1e-5, (n, rel_err(gb[n], ga[n]))


@pytest.mark.parametrize("S,B,D", [(3, 16, 90), (14, 50, 90), (2, 7, 12), (11, 16, 90), (3, 40, 90)])
def test_gp_elbo_steps_equals_the_per_step_loop(S, B, D):
    """gp_autograd.gp_elbo_steps (the GP posterior + ELBO term of S teacher-forced steps as ONE forward and ONE backward
    launch: S x D virtual latent dims with the parameters tiled S times) against the loop it replaces,
    `mll(gp_layer(h_i), h_target_i)` per step (train.py:164-169,225-226): the ELBO per (step, dim), the posterior means, and
    every gradient - inducing points, variational mean / Cholesky factor, constant mean, raw output scale / length scale,
    raw noise, and the input codes - under a weighting that differs per step and per dim (the closure-level
    test only saw this through a whole-arena norm)."""
    from dvg_amd.gp_autograd import gp_elbo_steps
    from dvg_amd.models import gp_models as gm
    M = 40
    sd, lik = params.gp_state(540, D=D, M=M)
    hin = params.normal(541, S, B, D, scale=0.7).tanh()
    htgt = params.normal(542, S, B, D, scale=0.5)
    w_elbo = params.normal(543, S, D).to(dev())
    w_mean = params.normal(544, S, B, D).to(dev())
    res = {}
    for batched in (True, False):
        gp, like = gm.GPRegressionLayer1(D, M), gm.GaussianLikelihood(batch_size=D)
        gp.load_state_dict(sd)
        like.load_state_dict(lik)
        gp.to(dev()).train(), like.to(dev()).train()
        mll = gm.VariationalELBO(like, gp, num_data=37)
        hi = hin.to(dev()).requires_grad_(True)
        ht = htgt.to(dev())
        if batched:
            elbo, mean = gp_elbo_steps(gp, mll, hi, ht)
            elbo = elbo.view(S, D)
        else:
            es, ms = [], []
            for i in range(S):
                pred = gp(hi[i].transpose(0, 1).view(D, B, 1))
                es.append(mll(pred, ht[i].transpose(0, 1)))
                ms.append(pred.mean.transpose(0, 1))
            elbo, mean = torch.stack(es), torch.stack(ms)
        ((elbo * w_elbo).sum() + (mean * w_mean).sum()).backward()
        g = {k: p.grad.clone() for k, p in gp.named_parameters()}
        g.update(h=hi.grad.clone(), noise=like.noise_covar.raw_noise.grad.clone())
        res[batched] = (elbo.detach().clone(), mean.detach().clone(), g)
    (ea, ma, ga), (eb, mb, gb) = res[True], res[False]
    assert ea.shape == (S, D) and ma.shape == (S, B, D)
    assert torch.equal(ma, mb), "same kernels, same arithmetic per (step, dim)"
    assert rel_err(ea, eb) < 1e-6       # the KL term is summed in fp64 by another number of threads, then rounded to fp32
    for k in gb:
        assert rel_err(ga[k], gb[k]) < 1e-5, (k, rel_err(ga[k], gb[k]))   # S gradient copies summed in a different order


@pytest.mark.parametrize("family", ["dcgan", "vgg"])
def test_in_place_parameter_gradients_equal_autograd_accumulation(family):
    """autograd.DIRECT_PARAM_GRADS: the backward kernels add parameter gradients straight into `.grad` (None returned to
    autograd).  Same gradients as handing them back to autograd, also when `.grad` already holds something (a second
    backward accumulates), when it starts as None, and through the shared-skip-half / upsample-as-transposed-conv paths."""
    from dvg_amd import autograd as ag
    from dvg_amd import fused
    mod = our_module(family, 64)
    x = params.frames(2100, 4, 1, 64).to(dev())
    gy = params.normal(2101, 4, 1, 64, 64).to(dev())
    res = {}
    for direct in (True, False, "batch3"):
        ag.DIRECT_PARAM_GRADS = bool(direct)
        old_batch, ag.WGRAD_BATCH = ag.WGRAD_BATCH, (3 if direct == "batch3" else (8 if direct else 1))
        try:
            enc, dec = mod.encoder(90, 1), mod.decoder(90, 1)
            enc.load_state_dict(params.fill_state_dict(enc.state_dict(), 2110))
            dec.load_state_dict(params.fill_state_dict(dec.state_dict(), 2111, params.decoder_transposed_keys(dec.state_dict(), family)))
            enc.to(dev()).train(), dec.to(dev()).train()
            for rep in range(2):                      # second pass accumulates on top of the first
                h, skips = enc(x)
                with fused.share_skip_halves():       # two decoder calls share the skip halves (train.py:227-231)
                    y = dec([h, skips]) + dec([h * 0.5, skips])
                ((y * gy).sum() + h.sum()).backward()
            res[direct] = {k: p.grad.detach().clone() for m_ in (enc, dec) for k, p in m_.named_parameters()}
            assert all(p.grad is not None for m_ in (enc, dec) for p in m_.parameters())
            assert not ag._wgrad_queues and not ag._dense_queues, "every queued weight gradient is flushed when backward() returns"
        finally:
            ag.DIRECT_PARAM_GRADS, ag.WGRAD_BATCH = True, old_batch
    for k, g in res[False].items():
        scale = max(float(g.abs().max()), 1e-20)
        assert float((res[True][k] - g).abs().max()) <= 2e-5 * scale + 1e-9, k          # batches of up to 8 uses per launch
        assert float((res["batch3"][k] - g).abs().max()) <= 2e-5 * scale + 1e-9, k      # batches of 3 (+ a remainder)


def test_in_place_gradients_lstm_linear():
    from dvg_amd import autograd as ag
    import dvg_amd.models.lstm as ours
    res = {}
    old_dense = ag.DENSE_BATCH
    for direct in (True, False, "per_use", "batch2"):   # default: one dW GEMM over all time steps of a backward pass
        ag.DIRECT_PARAM_GRADS = direct is not False
        ag.DENSE_BATCH = {"per_use": 1, "batch2": 2}.get(direct, old_dense)
        try:
            net = ours.lstm(90, 90, 256, 2, 6)
            net.load_state_dict(params.fill_state_dict(net.state_dict(), 2200))
            net.to(dev())
            for rep in range(2):
                net.hidden = net.init_hidden()
                loss = sum((net(params.normal(2210 + t, 6, 90, scale=0.5).to(dev())) ** 2).sum() for t in range(3))
                loss.backward()
            res[direct] = {k: p.grad.detach().clone() for k, p in net.named_parameters()}
            assert not ag._dense_queues
        finally:
            ag.DIRECT_PARAM_GRADS, ag.DENSE_BATCH = True, old_dense
    for k, g in res[False].items():
        for mode in (True, "per_use", "batch2"):
            assert float((res[mode][k] - g).abs().max()) <= 2e-5 * max(float(g.abs().max()), 1e-20) + 1e-9, (k, mode)
