"""GPU parity tests of the Winograd F(4x4,3x3) / F(2x2,3x3) forms and of the kernels that hand the activation from one Winograd
layer to the next (same resolution, through the max-pool, through the upsampling, from the decoder stem) against the direct form,
fp64 and the oracle.  Split out of tests/test_gpu_parity.py in r06 (no file above 800 lines)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dvg_oracle as orc
from oracle import params
from tests.common import BACKBONE_CASES, backbone_case, dev, nhwc, oracle_backbone, rel_err, summarize, to64, yardstick

pytestmark = pytest.mark.gpu
REL = 1e-4


@pytest.mark.parametrize("N,H,C1,Cout", [(32, 4, 512, 512), (8, 8, 256, 256), (2, 16, 128, 128), (64, 8, 256, 256)])
def test_upsample_conv3x3_winograd_with_hoisted_skip_half(N, H, C1, Cout):
    """The x half of a decoder block's first conv (vgg_64.py:93,98-105) in Winograd F(4x4,3x3) form: the input transform
    reads x through the nearest-x2 upsampling (dvg_winograd_input(upsample=1)), the hoisted skip half S enters the output
    transform as `addend`: y = act((conv3x3(up2(x), W_x) + S) * scale + shift) against the fp64 reference and against the
    transposed-conv (K4) form it replaces; handed over to the next layer (to_v) it must give that layer bit-identical results."""
    from dvg_amd import ops
    x = params.normal(3000, N, C1, H, H)
    w = params.normal(3001, Cout, C1, 3, 3, scale=1.2 / (3 * C1 ** 0.5))
    w2 = params.normal(3002, Cout, Cout, 3, 3, scale=1.2 / (3 * Cout ** 0.5))
    sc, sh = 1 + 0.1 * params.normal(3003, Cout), 0.1 * params.normal(3004, Cout)
    S = params.normal(3005, N, Cout, 2 * H, 2 * H, scale=0.3)
    d = lambda t: t.to(dev())   # noqa: E731
    up = F.interpolate(x, scale_factor=2, mode="nearest").double()
    ref = F.leaky_relu((F.conv2d(up, w.double(), padding=1) + S.double()) * sc.double().view(1, -1, 1, 1) +
                       sh.double().view(1, -1, 1, 1), 0.2)
    assert ops.winograd_ok(N, C1, 2 * H, 2 * H, Cout, 4)
    u = ops.winograd_weight(d(w), 4)
    y = ops.conv3x3_winograd(nhwc(x), u, d(sc), d(sh), upsample=True, addend=nhwc(S))
    assert y.shape == (N, Cout, 2 * H, 2 * H) and rel_err(y, ref) < 1e-4, rel_err(y, ref)
    # the form it replaces (fused._upconv_packed): same result up to fp32 summation order
    k4 = torch.zeros((Cout, C1, 4, 4))
    for ty in range(3):
        for tx in range(3):
            k4[:, :, 2 - ty:4 - ty, 2 - tx:4 - tx] += w[:, :, ty:ty + 1, tx:tx + 1]
    yk = ops.convT4x4s2(nhwc(x), None, ops.pack_igemm_weight(d(k4.permute(1, 0, 2, 3).contiguous()), transposed=True), d(sc),
                        d(sh), addend=nhwc(S))
    assert rel_err(y, yk) < 5e-5
    if ops.winograd_chain_ok(N, Cout, 2 * H, 2 * H):
        u2 = ops.winograd_weight(d(w2), 4)
        v = ops.conv3x3_winograd(nhwc(x), u, d(sc), d(sh), upsample=True, addend=nhwc(S), to_v=True)
        assert isinstance(v, ops.WinoV)
        assert torch.equal(ops.conv3x3_winograd(v, u2, d(sc), d(sh)), ops.conv3x3_winograd(y, u2, d(sc), d(sh)))


@pytest.mark.parametrize("N,H,C,Cout,pool", [(8, 8, 64, 64, False), (8, 8, 512, 256, True), (8, 16, 256, 256, False),
                                             (64, 8, 256, 512, True), (2, 32, 128, 64, False), (32, 8, 512, 512, True),
                                             (96, 32, 128, 128, True)])
def test_winograd_conv3x3_matches_direct_and_fp64(N, H, C, Cout, pool):
    """Winograd F(2x2,3x3) path (input transform -> 16 batched GEMMs in the igemm kernel's GEMM mode -> output transform with
    scale / shift / activation / 2x2 max-pool) against the fp64 reference and the direct implicit-GEMM kernel.  r06: cases with
    Cout % 128 == 0 and >= 256 workgroups of it run their F(4x4) GEMMs on the 128 x 128 tile of the bf16-triple build (64 x 64 per
    wave, K = 32 per stage, LEAN fragments, two workgroups per CU: (64, 8, 256, 512), (32, 8, 512, 512), (96, 32, 128, 128)), the
    others on the 64 x 64 tile; the f32-MFMA build has the 64- / 128-row tiles only."""
    from dvg_amd import ops
    x = params.normal(2300, N, C, H, H)
    w = params.normal(2301, Cout, C, 3, 3, scale=1.2 / (3 * C ** 0.5))
    sc, sh = 1 + 0.1 * params.normal(2302, Cout), 0.1 * params.normal(2303, Cout)
    ref = F.leaky_relu(F.conv2d(x.double(), w.double(), padding=1) * sc.double().view(1, -1, 1, 1) +
                       sh.double().view(1, -1, 1, 1), 0.2)
    wd = w.to(dev())
    assert ops.winograd_ok(N, C, H, H, Cout)
    direct = ops.conv3x3(nhwc(x), None, ops.pack_igemm_weight(wd), sc.to(dev()), sh.to(dev()), pool=pool)
    yd = direct[0] if pool else direct
    for m, tol in ((2, 1e-5), (4, 4e-5)):      # F(4x4,3x3) rounds ~5x coarser than F(2x2,3x3) (its transforms scale by up to 8)
        if not ops.winograd_ok(N, C, H, H, Cout, m):
            assert m == 4
            continue
        out = ops.conv3x3_winograd(nhwc(x), ops.winograd_weight(wd, m), sc.to(dev()), sh.to(dev()), pool=pool)
        y = out[0] if pool else out
        assert rel_err(y, ref) < tol, (m, rel_err(y, ref))
        assert rel_err(y, yd) < tol
        if pool:
            assert rel_err(out[1], F.max_pool2d(ref, 2, 2)) < tol
            assert torch.equal(out[1], F.max_pool2d(out[0], 2, 2)), "the pooled output is the max of the stored outputs, bit for bit"
    assert not ops.winograd_ok(4, 512, 8, 8, 512)     # 64 tiles: not a whole GEMM tile -> the caller keeps the direct kernel


@pytest.mark.parametrize("family,B", [("vgg", 32), ("vgg", 8)])
def test_eval_backbone_at_winograd_batch_matches_oracle(family, B):
    """The golden cases run at B <= 4, where no layer has enough output tiles for the Winograd path; here the eval-mode
    encoder -> decoder runs at a batch where the deep 3x3 layers DO take it (B = 32: F(4x4) on 8x8 / 16x16 / 32x32 maps; B = 8:
    F(2x2) on 8x8, F(4x4) above) and must still match the oracle (pinned to the reference) within the 1e-4 bar."""
    from dvg_amd import fused, ops
    enc, dec, esd, dsd, _, _ = backbone_case("vgg_64/eval")
    x = params.frames(2400, B, 1, 64)
    with torch.no_grad():
        h_ref, skips_ref = orc.vgg_encoder(x, esd, False)
        y_ref = orc.vgg_decoder(h_ref, skips_ref, dsd, False)
    enc.to(dev()).eval(), dec.to(dev()).eval()
    used = []
    real = ops.conv3x3_winograd
    ops.conv3x3_winograd = lambda xx, u, *a, **k: (used.append((u.shape[0], tuple(xx.shape))), real(xx, u, *a, **k))[1]
    try:
        with torch.no_grad():
            h, skips = enc(x.to(dev()))
            y = dec([h, skips])
    finally:
        ops.conv3x3_winograd = real
    if fused.WINOGRAD == 4:      # (under DVG_WINOGRAD=0 / 2 the same parity bars apply to whatever path runs)
        assert len(used) >= 8 and (36 in {u for u, _ in used}), used
        if B == 8:
            assert 16 in {u for u, _ in used}, "8x8 maps at B = 8 have 128 F(2x2) tiles but only 32 F(4x4) tiles"
    e_h, e_y = rel_err(h, h_ref), rel_err(y, y_ref)
    print(f"winograd backbone B={B}: rel err latent {e_h:.2e} frame {e_y:.2e} ({len(used)} winograd layers)")
    assert e_h < 1e-4 and e_y < 1e-4
    for a, b in zip(skips, skips_ref):
        assert rel_err(a, b) < 1e-4


@pytest.mark.parametrize("N,H,C,Cmid,Cout", [(32, 8, 256, 512, 512), (8, 16, 128, 256, 256), (32, 8, 512, 512, 256),
                                             (8, 32, 64, 128, 128), (32, 16, 128, 256, 256)])
def test_winograd_chain_hands_over_the_input_transform(N, H, C, Cmid, Cout):
    """dvg_winograd_output_input: two consecutive F(4x4,3x3) layers with the first layer's activation never written - the
    second layer's result equals (bit for bit: same kernels around it, same arithmetic inside) the unchained pair's, and both
    match the fp64 reference."""
    from dvg_amd import ops
    x = params.normal(2500, N, C, H, H)
    w1 = params.normal(2501, Cmid, C, 3, 3, scale=1.2 / (3 * C ** 0.5))
    w2 = params.normal(2502, Cout, Cmid, 3, 3, scale=1.2 / (3 * Cmid ** 0.5))
    s1, b1 = 1 + 0.1 * params.normal(2503, Cmid), 0.1 * params.normal(2504, Cmid)
    s2, b2 = 1 + 0.1 * params.normal(2505, Cout), 0.1 * params.normal(2506, Cout)
    mid = F.leaky_relu(F.conv2d(x.double(), w1.double(), padding=1) * s1.double().view(1, -1, 1, 1) + b1.double().view(1, -1, 1, 1), 0.2)
    ref = F.leaky_relu(F.conv2d(mid, w2.double(), padding=1) * s2.double().view(1, -1, 1, 1) + b2.double().view(1, -1, 1, 1), 0.2)
    u1, u2 = ops.winograd_weight(w1.to(dev()), 4), ops.winograd_weight(w2.to(dev()), 4)
    d = lambda t: t.to(dev())   # noqa: E731
    y1 = ops.conv3x3_winograd(nhwc(x), u1, d(s1), d(b1))
    y2 = ops.conv3x3_winograd(y1, u2, d(s2), d(b2))
    assert ops.winograd_chain_ok(N, Cmid, H, H)
    v = ops.conv3x3_winograd(nhwc(x), u1, d(s1), d(b1), to_v=True)
    assert isinstance(v, ops.WinoV) and v.shape == (N, Cmid, H, H)
    y2c = ops.conv3x3_winograd(v, u2, d(s2), d(b2))
    assert torch.equal(y2c, y2)
    assert rel_err(y2c, ref) < 1e-4
    # three in a row, the last one pooled
    v2 = ops.conv3x3_winograd(v, u2, d(s2), d(b2), to_v=True) if Cout == Cmid else None
    if v2 is not None:
        a = ops.conv3x3_winograd(v2, u2, d(s2), d(b2), pool=True)
        b = ops.conv3x3_winograd(y2, u2, d(s2), d(b2), pool=True)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        # ... and the pooled map handed over to the next STAGE as its input transform (dvg_winograd_output_pool_input):
        # the skip tensor and the next stage's first layer must come out bit-identical to the unchained route
        if ops.winograd_pool_chain_ok(N, Cout, H, H) and ops.winograd_ok(N, Cout, H // 2, H // 2, Cout, 4):
            ysk, vp = ops.conv3x3_winograd(v2, u2, d(s2), d(b2), pool=True, to_v=True)
            assert isinstance(vp, ops.WinoV) and vp.shape == (N, Cout, H // 2, H // 2)
            assert torch.equal(ysk, b[0])
            w3 = params.normal(2507, Cout, Cout, 3, 3, scale=1.2 / (3 * Cout ** 0.5))
            u3 = ops.winograd_weight(w3.to(dev()), 4)
            assert torch.equal(ops.conv3x3_winograd(vp, u3, d(s2), d(b2)), ops.conv3x3_winograd(b[1], u3, d(s2), d(b2)))


def test_eval_rollout_modules_chain_equals_unchained():
    """vgg_64 encoder -> decoder in eval mode at a Winograd batch with and without the WinoV hand-over (fused.WINOGRAD_CHAIN):
    identical outputs, and the chained run launches dvg_winograd_output_input."""
    from dvg_amd import fused, ops
    enc, dec, esd, dsd, _, _ = backbone_case("vgg_64/eval")
    x = params.frames(2600, 32, 1, 64).to(dev())
    enc.to(dev()).eval(), dec.to(dev()).eval()
    out = {}
    for chain in (True, False):
        old = fused.WINOGRAD_CHAIN
        fused.WINOGRAD_CHAIN = chain
        timer = ops.KernelTimer()
        ops.set_timer(timer)
        try:
            with torch.no_grad():
                h, skips = enc(x)
                y = dec([h, skips])
        finally:
            ops.set_timer(None)
            fused.WINOGRAD_CHAIN = old
        out[chain] = (h, skips, y, timer.summary())
    n_fused = out[True][3].get("winograd_output_input", {}).get("launches", 0)
    n_pool = out[True][3].get("winograd_output_pool_input", {}).get("launches", 0)
    assert "winograd_output_input" not in out[False][3] and "winograd_output_pool_input" not in out[False][3], list(out[False][3])
    if fused.WINOGRAD == 4:
        assert n_fused >= 5, n_fused
        if fused._CHAIN_LEVEL >= 2:   # c2.0 -> c2.1 at 32x32 inside the block; c2 -> c3 and c3 -> c4 through the max-pool
            assert n_fused >= 7 and n_pool == 2, (n_fused, n_pool)   # 5 in the encoder (one of them at 32x32), 2 in the decoder
    assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][2], out[False][2])
    for a, b in zip(out[True][1], out[False][1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("N", [16, 50])
def test_first_stage_as_one_launch_matches_two_launches_and_fp64(N):
    """dvg_conv3x3_first_pair: vgg_64's c1 = vgg_layer(1, 64) -> vgg_layer(64, 64) + MaxPool (vgg_64.py:23-26, 49) in eval mode
    with the first layer's activation computed inside the second layer's kernel - against the two-launch path (same kernels'
    arithmetic otherwise) and an fp64 torch composition, skip tensor and pooled map, at a batch that is not a multiple of 8."""
    import torch.nn as nn
    from dvg_amd import fused, ops
    g = torch.Generator().manual_seed(4100 + N)
    conv0, bn0, conv1, bn1 = nn.Conv2d(1, 64, 3, 1, 1), nn.BatchNorm2d(64), nn.Conv2d(64, 64, 3, 1, 1), nn.BatchNorm2d(64)
    with torch.no_grad():
        for conv, bn, fan in ((conv0, bn0, 9), (conv1, bn1, 576)):
            conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / fan) ** 0.5)
            conv.bias.copy_(torch.randn(conv.bias.shape, generator=g) * 0.1)
            bn.weight.copy_(1 + 0.2 * torch.randn(64, generator=g))
            bn.bias.copy_(0.1 * torch.randn(64, generator=g))
            bn.running_mean.copy_(0.1 * torch.randn(64, generator=g))
            bn.running_var.copy_(0.5 + torch.rand(64, generator=g))
    x = torch.rand(N, 1, 64, 64, generator=g)
    mods = nn.Sequential(conv0, bn0, nn.LeakyReLU(0.2), conv1, bn1, nn.LeakyReLU(0.2)).double().eval()
    with torch.no_grad():
        ref = mods(x.double())
        ref_pool = F.max_pool2d(ref, 2, 2)
    for m in (conv0, bn0, conv1, bn1):
        m.float().to("cuda:0").eval()
    xd = x.to("cuda:0")
    with torch.no_grad():
        assert fused.first_pair_applies(conv0, bn0, conv1, bn1, xd) or not fused.FIRST_PAIR      # (FIRST_PAIR off: the encoder takes two launches; the op itself is tested either way)
        y, yp = fused.conv3_first_pair(conv0, bn0, conv1, bn1, xd, pool=True)
        h0 = fused.conv3_first_bn_act(conv0, bn0, xd)
        y2, yp2 = fused.conv3_bn_act(conv1, bn1, h0, pool=True)
    assert ops.is_nhwc(y) and y.shape == (N, 64, 64, 64) and yp.shape == (N, 64, 32, 32)
    assert rel_err(y, ref) < 1e-5 and rel_err(yp, ref_pool) < 1e-5, (rel_err(y, ref), rel_err(yp, ref_pool))
    assert rel_err(y, y2) < 5e-6 and rel_err(yp, yp2) < 5e-6, (rel_err(y, y2), rel_err(yp, yp2))
    # pooled map == max-pool of the full map the same launch wrote, bit for bit
    assert torch.equal(yp, F.max_pool2d(y, 2, 2))
    # and the launch is deterministic (a first version selected the LeakyReLU branch and the zero padding with lane masks
    # inside the MFMA-interleaved store phase and produced run-to-run different tiles with two workgroups per CU)
    with torch.no_grad():
        for _ in range(10):
            assert torch.equal(fused.conv3_first_pair(conv0, bn0, conv1, bn1, xd, pool=True)[0], y)


@pytest.mark.parametrize("N,C,Cmid,Cout", [(32, 512, 256, 256), (64, 256, 128, 128)])
def test_winograd_chain_through_the_upsampling(N, C, Cmid, Cout):
    """dvg_winograd_output_up_input: the last layer of a decoder block (8 x 8) hands the input transform of its UPSAMPLED output to
    the x half of the next block's concat conv (vgg_64.py:93,98-105) - bit-identical to writing the activation and letting
    that conv transform it through the upsampling (dvg_winograd_output + dvg_winograd_input(upsample = 1)), and both match fp64."""
    from dvg_amd import ops
    H = 8
    x = params.normal(2800, N, C, H, H)
    w1 = params.normal(2801, Cmid, C, 3, 3, scale=1.2 / (3 * C ** 0.5))
    w2 = params.normal(2802, Cout, Cmid, 3, 3, scale=1.2 / (3 * Cmid ** 0.5))     # the x half of the next block's concat conv
    s1, b1 = 1 + 0.1 * params.normal(2803, Cmid), 0.1 * params.normal(2804, Cmid)
    s2, b2 = 1 + 0.1 * params.normal(2805, Cout), 0.1 * params.normal(2806, Cout)
    add = params.normal(2807, N, Cout, 2 * H, 2 * H, scale=0.3)                    # the hoisted skip half (raw sums)
    f64 = lambda t: t.double()   # noqa: E731
    mid = F.leaky_relu(F.conv2d(f64(x), f64(w1), padding=1) * f64(s1).view(1, -1, 1, 1) + f64(b1).view(1, -1, 1, 1), 0.2)
    up = F.interpolate(mid, scale_factor=2, mode="nearest")
    ref = F.leaky_relu((F.conv2d(up, f64(w2), padding=1) + f64(add)) * f64(s2).view(1, -1, 1, 1) + f64(b2).view(1, -1, 1, 1), 0.2)
    d = lambda t: t.to(dev())   # noqa: E731
    u1, u2 = ops.winograd_weight(d(w1), 4), ops.winograd_weight(d(w2), 4)
    assert ops.winograd_up_chain_ok(N, Cmid, H, H)
    y1 = ops.conv3x3_winograd(nhwc(x), u1, d(s1), d(b1))
    y2 = ops.conv3x3_winograd(y1, u2, d(s2), d(b2), upsample=True, addend=nhwc(add))
    v = ops.conv3x3_winograd(nhwc(x), u1, d(s1), d(b1), to_v="up")
    assert isinstance(v, ops.WinoV) and v.up and v.shape == (N, Cmid, 2 * H, 2 * H)
    y2c = ops.conv3x3_winograd(v, u2, d(s2), d(b2), upsample=True, addend=nhwc(add))
    assert torch.equal(y2c, y2)
    assert rel_err(y2c, ref) < 1e-4, rel_err(y2c, ref)
    with pytest.raises(RuntimeError):      # an upsampled WinoV is not a same-resolution input transform
        ops.conv3x3_winograd(v, u2, d(s2), d(b2))


@pytest.mark.parametrize("M,K", [(64, 90), (50, 90), (3, 128)])
def test_stem_hands_over_through_the_upsampling(M, K):
    """dvg_stem_up_winograd_input == dvg_stem_gemm followed by dvg_winograd_input(upsample = 1), bit for bit (ragged batch, both
    K paddings), and the stem map it stands for matches an fp64 composition."""
    from dvg_amd import ops
    C = 512
    KP = 96 if K <= 96 else 128
    vec = params.normal(2820, M, K, scale=0.5).tanh().to(dev())
    w = params.normal(2821, K, C, 4, 4, scale=0.05)                      # ConvTranspose2d(K, C, 4, 1, 0).weight
    wt = torch.zeros(KP, 16 * C)
    wt[:K] = w.permute(0, 2, 3, 1).reshape(K, 16 * C)
    wt = wt.to(dev())
    sc, sh = (1 + 0.1 * params.normal(2822, C)).to(dev()), (0.1 * params.normal(2823, C)).to(dev())
    out = ops.nhwc_empty(M, C, 4, 4, dev())
    ops.stem_gemm(vec, wt, K, sc, sh, out.permute(0, 2, 3, 1).reshape(M, 16 * C), period=C)
    ref = F.leaky_relu(torch.einsum("mk,kchw->mchw", vec.double().cpu(), w.double()) * sc.double().cpu().view(1, -1, 1, 1)
                       + sh.double().cpu().view(1, -1, 1, 1), 0.2)
    assert rel_err(out, ref) < 1e-5
    v_ref = torch.empty(36, 4 * M, C, device=dev())
    from dvg_amd._lib import check, lib
    check(lib().dvg_winograd_input(out.data_ptr(), v_ref.data_ptr(), M, 8, 8, C, 4, 1, torch.cuda.current_stream().cuda_stream), "in")
    wv = ops.stem_up_winograd_input(vec, wt, K, sc, sh, C)
    assert wv.up and wv.shape == (M, C, 8, 8) and torch.equal(wv.v, v_ref)


def test_decoder_blocks_hand_over_through_the_upsampling():
    """vgg_64 decoder in eval mode with frozen skips (a rollout's prediction steps): with DVG_WINOGRAD_CHAIN >= 3 the last layer of
    upc2 hands upc3's first conv its input transform through `up` (one launch instead of dvg_winograd_output +
    dvg_winograd_input), and the stem hands upc2's first conv its own (dvg_stem_up_winograd_input instead of dvg_stem_gemm +
    dvg_winograd_input); frames bit-identical to the level-2 run."""
    from dvg_amd import fused, ops
    enc, dec, esd, dsd, _, _ = backbone_case("vgg_64/eval")
    enc.to(dev()).eval(), dec.to(dev()).eval()
    x = params.frames(2810, 32, 1, 64).to(dev())
    out = {}
    for level in (3, 2):
        old = fused._CHAIN_LEVEL
        fused._CHAIN_LEVEL = level
        fused.clear_skip_hoist_cache()
        ops.clear_skip_proj_cache()
        try:
            with torch.no_grad():
                h, skips = enc(x)
                fused.declare_frozen_skips(skips)
                dec([h, skips])                     # (the first call computes the hoisted skip halves)
                timer = ops.KernelTimer()
                ops.set_timer(timer)
                y = dec([h, skips])
                ops.set_timer(None)
        finally:
            ops.set_timer(None)
            fused._CHAIN_LEVEL = old
        out[level] = (y, timer.summary())
    fused.clear_skip_hoist_cache()
    ops.clear_skip_proj_cache()
    assert torch.equal(out[3][0], out[2][0])
    if fused.WINOGRAD == 4 and fused.WINOGRAD_CHAIN and fused.SKIP_HOIST and fused.UPCONV_WINOGRAD:
        assert out[3][1].get("winograd_output_up_input", {}).get("launches", 0) == 1, list(out[3][1])
        assert out[3][1].get("stem_up_winograd_input", {}).get("launches", 0) == 1, list(out[3][1])    # stem -> upc2 likewise
        assert "winograd_output_up_input" not in out[2][1] and "stem_up_winograd_input" not in out[2][1]
        n3 = sum(v["launches"] for v in out[3][1].values())
        n2 = sum(v["launches"] for v in out[2][1].values())
        assert n3 == n2 - 2, (n3, n2)
