"""GPU: launch-to-launch determinism of every conv_igemm2 instantiation the configs use.

The r03 finding at dvg_conv3x3_first_pair - a store phase written with selects gave run-to-run different tiles with two
workgroups per CU - was root-caused in r04 (tools/ubench/): hipcc lowers such selects to EXEC-masked
basic blocks inside the MFMA-interleaved stage loop.  tests/test_isa_invariants.py proves on the shipped ISA that no other
instantiation has an EXEC change inside its stage loop; this file is the dynamic counterpart: every instantiation - the two
3x3 tiles (plain, with skip + upsample + addend: the `gload_a` address-select path on halo and out-of-image slots), the
FIRST pair, both stride-2 conv tiles, the three transposed-conv tiles, both Winograd batched-GEMM tiles (through
conv3x3_winograd) - launched REPS times with THREE chains in flight on three streams (co-resident workgroups of different
launches share the CUs, at >= 2 workgroups per CU) and compared BIT FOR BIT, element by element, with the first result."""
import pytest
import torch

from oracle import params

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
REPS = 50


def _aff(c, seed):
    return (1 + 0.1 * params.normal(seed, c)).to(DEV), (0.1 * params.normal(seed + 1, c)).to(DEV)


def _cases():
    """name -> (expected kernel tile tag, launch closure factory).  Shapes are the configs' own layers at B = 16 ... 64."""
    from dvg_amd import ops

    def nhwc(seed, *shape, scale=1.0):
        return ops.to_nhwc(params.normal(seed, *shape, scale=scale).to(DEV))

    cases = {}

    def conv3(n, c1, c2, hw, cout, up, pool, add, seed):
        hx = hw // 2 if up else hw
        x = nhwc(seed, n, c1, hx, hx)
        sk = nhwc(seed + 1, n, c2, hw, hw) if c2 else None
        wp = ops.pack_igemm_weight(params.normal(seed + 2, cout, c1 + c2, 3, 3, scale=0.05).to(DEV))
        sc, sh = _aff(cout, seed + 3)
        ad = nhwc(seed + 5, n, cout, hw, hw, scale=0.3) if add else None
        return lambda: ops.conv3x3(x, sk, wp, sc, sh, upsample=up, pool=pool, addend=ad)
    cases["conv3 8x16 plain 64x64"] = conv3(16, 64, 0, 64, 64, False, True, False, 5000)
    cases["conv3 8x16 up+skip 64x64"] = conv3(16, 64, 64, 64, 64, True, False, False, 5010)
    cases["conv3 8x16 up+addend 32x32"] = conv3(32, 128, 0, 32, 128, True, False, True, 5020)
    cases["conv3 8x8 16x16 256->256"] = conv3(64, 256, 0, 16, 256, False, True, False, 5030)
    cases["conv3 8x8 up+skip 8x8 (split-K)"] = conv3(16, 512, 512, 8, 512, True, False, False, 5040)

    def first_pair(n, seed):
        import torch.nn as nn
        from dvg_amd import fused
        g = torch.Generator().manual_seed(seed)
        mods = [nn.Conv2d(1, 64, 3, 1, 1), nn.BatchNorm2d(64), nn.Conv2d(64, 64, 3, 1, 1), nn.BatchNorm2d(64)]
        with torch.no_grad():
            for bn in (mods[1], mods[3]):
                bn.running_mean.copy_(0.1 * torch.randn(64, generator=g))
                bn.running_var.copy_(0.5 + torch.rand(64, generator=g))
        for m in mods:
            m.to(DEV).eval()
        x = torch.rand(n, 1, 64, 64, generator=g).to(DEV)
        return lambda: fused.conv3_first_pair(*mods, x, pool=True)
    cases["conv3 FIRST pair 64x64"] = first_pair(16, 5050)

    def conv4s2(n, cin, hw, cout, seed):
        x = nhwc(seed, n, cin, hw, hw)
        wp = ops.pack_igemm_weight(params.normal(seed + 1, cout, cin, 4, 4, scale=0.05).to(DEV))
        sc, sh = _aff(cout, seed + 2)
        return lambda: ops.conv4x4s2(x, wp, sc, sh)
    cases["conv4s2 8x8 32x32 64->128"] = conv4s2(64, 64, 32, 128, 5060)
    cases["conv4s2 4x4x4 8x8 256->512"] = conv4s2(64, 256, 8, 512, 5070)

    def convT(n, c1, c2, hw, cout, add, seed):
        x = nhwc(seed, n, c1, hw, hw)
        sk = nhwc(seed + 1, n, c2, hw, hw) if c2 else None
        wp = ops.pack_igemm_weight(params.normal(seed + 2, c1 + c2, cout, 4, 4, scale=0.05).to(DEV), transposed=True)
        sc, sh = _aff(cout, seed + 3)
        ad = nhwc(seed + 5, n, cout, 2 * hw, 2 * hw, scale=0.3) if add else None
        return lambda: ops.convT4x4s2(x, sk, wp, sc, sh, addend=ad)
    cases["convT 8x16 16x16 128+128->64"] = convT(64, 128, 128, 16, 64, False, 5080)
    cases["convT 8x8 8x8 256->128 +addend"] = convT(64, 256, 0, 8, 128, True, 5090)
    cases["convT 4x4x4 4x4 512+512->256"] = convT(64, 512, 512, 4, 256, False, 5100)

    def wino(n, c, hw, cout, seed):
        x = nhwc(seed, n, c, hw, hw)
        u = ops.winograd_weight(params.normal(seed + 1, cout, c, 3, 3, scale=0.05).to(DEV), 4)
        sc, sh = _aff(cout, seed + 2)
        return lambda: ops.conv3x3_winograd(x, u, sc, sh)
    cases["winograd GEMM 16x16 256->256 B=64"] = wino(64, 256, 16, 256, 5110)
    cases["winograd GEMM 8x8 512->512 B=64"] = wino(64, 512, 8, 512, 5120)
    cases["winograd GEMM 32x32 128->128 B=64"] = wino(64, 128, 32, 128, 5130)
    cases["winograd GEMM 8x8 512->512 B=576"] = wino(576, 512, 8, 512, 5140)
    return cases


def _flat(out):
    return [t for t in (out if isinstance(out, (tuple, list)) else (out,)) if torch.is_tensor(t)]


def test_every_conv_igemm2_instantiation_is_bit_deterministic_with_three_chains_in_flight():
    from dvg_amd import ops
    cases = _cases()
    timer = ops.KernelTimer()
    streams = [torch.cuda.Stream() for _ in range(3)]
    bad = []
    with torch.no_grad():
        for name, launch in cases.items():
            first = [t.clone() for t in _flat(launch())]
            torch.cuda.synchronize()
            assert all(bool(torch.isfinite(t).all()) for t in first), name
            ndiff = worst = 0
            for rep in range(0, REPS, 3):
                outs = []
                for s in streams:        # three launches of the SAME op in flight at once, each into fresh output buffers
                    s.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(s):
                        outs.append(_flat(launch()))
                for s in streams:
                    torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                for o in outs:
                    for a, b in zip(o, first):
                        d = int((a != b).sum())          # element-wise, not a tensor-wide error figure
                        if d:
                            ndiff += 1
                            worst = max(worst, d)
            if ndiff:
                bad.append((name, ndiff, worst))
    assert not bad, f"launches that differ from the first one (name, launches, worst element count): {bad}"
    del timer


def test_rollout_chains_in_flight_reproduce_the_single_chain_bit_for_bit():
    """The bench's own regime: three complete vgg_64 / dcgan_64 rollouts in flight (rollout.ConcurrentRollouts, one hipGraph and
    stream each) against the same rollout as one serial chain - every predicted frame before the GP trigger step (whose eps is
    fresh per replay) bit-identical, over several rounds."""
    from dvg_amd.rollout import ConcurrentRollouts
    from tests.test_gpu_configs import _build
    B, n_past, n_eval = 64, 10, 20
    for family in ("vgg", "dcgan"):
        mods, _ = _build(family, 64, 1, B, 5200)
        for m in mods:
            m.to(DEV).eval()
        xs = [params.frames(5210 + t, B, 1, 64).to(DEV) for t in range(n_eval)]
        cr = ConcurrentRollouts(*mods, xs, n_past, n_eval, inflight=3)
        ref = [f.clone() for f in cr.run(1, chains=1)[0]]
        for _ in range(4):
            outs = cr.run(3)
            torch.cuda.synchronize()
            for o in outs:
                for t in range(n_past, 15):
                    assert torch.equal(o[t], ref[t]), (family, t, int((o[t] != ref[t]).sum()))


def test_splitk_launches_on_three_streams_are_bit_identical():
    """Split K (partial tiles to a per-stream workspace + splitk_finish_kernel, fixed summation order): every output bit for bit
    the same - with pool, with a raw addend, with fused upsample + skip, on 4x4x4-image tiles with a ragged last tile, on the
    transposed conv's four parities - over REPS launches on three streams at once (each stream has its own workspace)."""
    from dvg_amd import ops
    from dvg_amd._lib import lib

    def nhwc(seed, *shape, scale=1.0):
        return ops.to_nhwc(params.normal(seed, *shape, scale=scale).to(DEV))

    def conv3(n, c1, c2, hw, cout, up, pool, add, seed):
        hx = hw // 2 if up else hw
        x = nhwc(seed, n, c1, hx, hx)
        sk = nhwc(seed + 1, n, c2, hw, hw) if c2 else None
        wp = ops.pack_igemm_weight(params.normal(seed + 2, cout, c1 + c2, 3, 3, scale=0.05).to(DEV))
        sc, sh = _aff(cout, seed + 3)
        ad = nhwc(seed + 5, n, cout, hw, hw, scale=0.3) if add else None
        assert lib().dvg_conv_splitk_v2(ops.MODE_CONV3, n, hw, hw, c1 + c2, cout) > 1
        return lambda: ops.conv3x3(x, sk, wp, sc, sh, upsample=up, pool=pool, addend=ad)

    def conv4s2(n, cin, hw, cout, seed):
        x = nhwc(seed, n, cin, hw, hw)
        wp = ops.pack_igemm_weight(params.normal(seed + 1, cout, cin, 4, 4, scale=0.05).to(DEV))
        sc, sh = _aff(cout, seed + 2)
        assert lib().dvg_conv_splitk_v2(ops.MODE_CONV4S2, n, hw, hw, cin, cout) > 1
        return lambda: ops.conv4x4s2(x, wp, sc, sh)

    def convT(n, c1, c2, hw, cout, add, seed):
        x = nhwc(seed, n, c1, hw, hw)
        sk = nhwc(seed + 1, n, c2, hw, hw) if c2 else None
        wp = ops.pack_igemm_weight(params.normal(seed + 2, c1 + c2, cout, 4, 4, scale=0.05).to(DEV), transposed=True)
        sc, sh = _aff(cout, seed + 3)
        ad = nhwc(seed + 5, n, cout, 2 * hw, 2 * hw, scale=0.3) if add else None
        assert lib().dvg_conv_splitk_v2(ops.MODE_CONVT4S2, n, hw, hw, c1 + c2, cout) > 1
        return lambda: ops.convT4x4s2(x, sk, wp, sc, sh, addend=ad)

    cases = {
        "conv3 8x8 up+skip 512+512->512": conv3(16, 512, 512, 8, 512, True, False, False, 5300),
        "conv3 8x8 512->256 pool (x8)": conv3(8, 512, 0, 8, 256, False, True, False, 5310),
        "conv3 16x16 128->128 addend": conv3(8, 128, 0, 16, 128, False, False, True, 5320),
        "conv4s2 16x16 128->256 (x2)": conv4s2(64, 128, 16, 256, 5330),
        "conv4s2 8x8 256->512 (x4)": conv4s2(64, 256, 8, 512, 5340),
        "conv4s2 8x8 256->512 B=6 (ragged 4-image tile)": conv4s2(6, 256, 8, 512, 5350),
        "convT 4x4 512+512->256": convT(64, 512, 512, 4, 256, False, 5360),
        "convT 8x8 256->128 +addend B=16": convT(16, 256, 0, 8, 128, True, 5370),
    }
    streams = [torch.cuda.Stream() for _ in range(3)]
    bad = []
    with torch.no_grad():
        for name, launch in cases.items():
            ref = [t.clone() for t in _flat(launch())]
            torch.cuda.synchronize()
            ndiff = 0
            for rep in range(0, REPS, 3):
                outs = []
                for s in streams:
                    s.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(s):
                        outs.append(_flat(launch()))
                for s in streams:
                    torch.cuda.current_stream().wait_stream(s)
                torch.cuda.synchronize()
                for o in outs:
                    ndiff += sum(int((a != b).sum()) for a, b in zip(o, ref))
            if ndiff:
                bad.append((name, ndiff))
    assert not bad, f"split-K launches differ between streams / repetitions (name, elements over all launches): {bad}"
    assert len({k[1] for k in ops._SPLITK_WS}) >= 3       # one workspace per stream
