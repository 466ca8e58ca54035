"""GPU parity of the TRAINING path: gradients produced by the HIP backward kernels (through
torch.autograd.Function glue) against torch autograd run on the CPU oracle for the same seeded
parameters and inputs.  Tolerance is relative to the largest gradient entry of each tensor."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dvg_oracle as orc
from oracle import params
from tests.common import BACKBONE_CASES, backbone_case, rel_err

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


def _oracle_grads(tag, family, dtype):
    enc, dec, esd, dsd, x, vec = backbone_case(tag)
    gy = params.normal(900, *x.shape).to(dtype)
    gh = params.normal(901, x.shape[0], 90).to(dtype)

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
    ((y * gy).sum() + (h * gh).sum()).backward()
    return h.detach(), y.detach(), e, d


def _module_grads(tag, family):
    """Gradients through LeakyReLU + batch-statistics BatchNorm are ill-conditioned at B=4 (an element
    whose pre-activation is within rounding of 0 flips its derivative between 1 and 0.2), so the yardstick
    is the fp64 oracle and the allowance is what the fp32 CPU oracle itself deviates from it."""
    h64, y64, e64, d64 = _oracle_grads(tag, family, torch.float64)
    h32, y32, e32, d32 = _oracle_grads(tag, family, torch.float32)
    enc, dec, esd, dsd, x, vec = backbone_case(tag)
    gy = params.normal(900, *x.shape)
    gh = params.normal(901, x.shape[0], 90)
    enc.to(dev()).train(), dec.to(dev()).train()
    ho, so = enc(x.to(dev()))
    yo = dec([ho, so])
    ((yo * gy.to(dev())).sum() + (ho * gh.to(dev())).sum()).backward()
    assert rel_err(yo, y64) < 5e-4 and rel_err(ho, h64) < 5e-4
    bad = []
    for name, r64, r32, ours in (("enc", e64, e32, dict(enc.named_parameters())),
                                 ("dec", d64, d32, dict(dec.named_parameters()))):
        for k, p in ours.items():
            g = r64[k].grad
            if g is None:
                continue
            if (k.endswith(".0.bias") and "main" in k) or k in ("c5.0.bias", "upc1.0.bias"):
                continue  # conv bias feeding a train-mode BatchNorm: analytically zero, noise on every side
            scale = max(float(g.abs().max()), 1e-12)
            diff = p.grad.double().cpu() - g
            err = float(diff.abs().max()) / scale
            l2 = float(diff.norm() / g.norm().clamp_min(1e-12))
            cdiff = r32[k].grad.double() - g
            cpu = float(cdiff.abs().max()) / scale
            cpu_l2 = float(cdiff.norm() / g.norm().clamp_min(1e-12))
            # a single kink flip moves a few entries by O(1e-2) of the max but barely moves the L2 norm; a wrong
            # kernel moves both by O(1).  The fp32 CPU oracle's own distance from fp64 is the yardstick.
            # (kernel-level gradient tests above hold 2e-3; this module-level check guards the WIRING, where a
            #  mistake shows up as O(1) — a dropped skip gradient, a wrong channel slice, a missing upsample sum)
            if not (l2 < max(2e-2, 3.0 * cpu_l2) and err < max(0.15, 3.0 * cpu)):
                bad.append((name, k, err, l2, cpu, cpu_l2))
    assert not bad, bad[:8]


def test_vgg64_module_backward():
    _module_grads("vgg_64/train", "vgg")


def test_dcgan64_module_backward():
    _module_grads("dcgan_64/train", "dcgan")


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
    assert abs(float(lo) - float(loss)) < 1e-3 * abs(float(loss)) + 1e-4
    grads_close({k: p.grad for k, p in net.named_parameters()}, {k: v.grad for k, v in ref.items()}, tol=1e-3)
    for t in range(4):
        assert rel_err(xo[t].grad, xr[t].grad) < 1e-3


@pytest.mark.parametrize("B,D,M", [(16, 12, 40), (64, 90, 40), (50, 7, 24)])
def test_gp_train_backward(B, D, M):
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
    assert abs(float(loss_o) - float(loss_r)) < 2e-3 * abs(float(loss_r)) + 1e-3
    ours = {k: p.grad for k, p in gp.named_parameters()}
    ours["h"] = ho.grad
    ours["noise"] = like.noise_covar.raw_noise.grad
    ref = {k: v.grad for k, v in rs.items()}
    ref["h"] = hr.grad
    ref["noise"] = rl["noise_covar.raw_noise"].grad
    ref["variational_strategy.variational_distribution.chol_variational_covar"] = torch.tril(
        ref["variational_strategy.variational_distribution.chol_variational_covar"])
    grads_close(ours, ref, tol=5e-3)
