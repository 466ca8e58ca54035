"""CPU: the arithmetic identity behind DVG_BF16X3 (dvg_amd/csrc/dvg_common.h), restated in numpy bit for bit.

The implicit-GEMM kernels split every fp32 operand into three bf16 terms by rounding to nearest even (v_cvt_pk_bf16_f32),
    h = bf16(a),  m = bf16(a - h),  l = bf16(a - h - m),
and form a product as the six terms (h,h) (h,m) (m,h) (h,l) (m,m) (l,h).  Checked here, on the same operations:
  * the split is EXACT (h + m + l == a in fp32, each term representable in bf16) for normal numbers whose third term is not
    subnormal (|a| > 2^-102; below that the last bits flush, 40 orders of magnitude under anything the networks hold);
  * bf16 x bf16 products are exact in fp32;
  * the three dropped terms are below 2^-24 |a b| with no preferred sign, i.e. below the rounding of one fp32 product (a split
    by truncation would leave up to 2^-20, always with the product's sign), and a K = 4608 dot product formed
    from the six terms (fp32 accumulation, as the MFMA does) is as close to the fp64 result as the plain fp32 one."""
import numpy as np

def bf16_rn(x: np.ndarray) -> np.ndarray:
    """fp32 -> nearest bf16 (ties to even), returned as fp32: what v_cvt_pk_bf16_f32 does for finite values."""
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    return ((u + ((u >> 16) & 1) + 0x7FFF) & 0xFFFF0000).astype(np.uint32).view(np.float32)


def split3(a: np.ndarray):
    a = a.astype(np.float32)
    h = bf16_rn(a)
    r = (a - h).astype(np.float32)
    m = bf16_rn(r)
    t = (r - m).astype(np.float32)
    return h, m, bf16_rn(t)


def is_bf16(x: np.ndarray) -> bool:
    return bool(np.all((x.astype(np.float32).view(np.uint32) & np.uint32(0xFFFF)) == 0))


def sample(n, rng, emax=60):
    mant = rng.standard_normal(n).astype(np.float32)
    expo = rng.integers(-emax, emax, n)
    return (mant * np.exp2(expo).astype(np.float32)).astype(np.float32)


def test_split_is_exact_and_every_term_is_a_bf16():
    rng = np.random.default_rng(0)
    a = np.concatenate([sample(200000, rng), np.float32([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 2.0 - 2.0 ** -23, 1.0e38, 2.0 ** -100,
                                                           0.1, -0.3, 16777215.0])])
    h, m, l = split3(a)
    assert is_bf16(h) and is_bf16(m) and is_bf16(l)
    # exact: the two subtractions are exact in fp32 and the last remainder has at most 8 significant bits
    assert np.array_equal((h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)), a.astype(np.float64))
    nz = a != 0
    assert np.all(np.abs(m[nz]) <= np.abs(a[nz]) * 2.0 ** -8) and np.all(np.abs(l[nz]) <= np.abs(a[nz]) * 2.0 ** -16)


def test_bf16_products_are_exact_in_fp32_and_the_dropped_terms_are_below_one_rounding():
    rng = np.random.default_rng(1)
    a, b = sample(100000, rng, 30), sample(100000, rng, 30)      # products of the smallest terms stay normal
    ah, am, al = split3(a)
    bh, bm, bl = split3(b)
    for x, y in ((ah, bh), (ah, bm), (am, bh), (ah, bl), (am, bm), (al, bh)):
        p32 = (x * y).astype(np.float32)
        assert np.array_equal(p32.astype(np.float64), x.astype(np.float64) * y.astype(np.float64))      # 8 x 8 bits fit 24
    kept = sum(x.astype(np.float64) * y.astype(np.float64) for x, y in ((ah, bh), (ah, bm), (am, bh), (ah, bl), (am, bm), (al, bh)))
    exact = a.astype(np.float64) * b.astype(np.float64)
    rel = (kept - exact) / np.abs(exact)
    assert np.abs(rel).max() <= 2.0 ** -24 and abs(rel.mean()) < 1e-10          # below one fp32 rounding, unbiased


def test_dot_product_from_six_terms_matches_fp64_like_plain_fp32():
    rng = np.random.default_rng(2)
    K, rows = 4608, 256
    a = rng.standard_normal((rows, K)).astype(np.float32)
    b = (rng.standard_normal((rows, K)) * (2.0 / K) ** 0.5).astype(np.float32)
    ref = np.einsum("rk,rk->r", a.astype(np.float64), b.astype(np.float64))
    ah, am, al = split3(a)
    bh, bm, bl = split3(b)
    acc = np.zeros(rows, np.float32)
    for k0 in range(0, K, 16):       # one "MFMA" per term and K = 16 slab: exact products, one fp32 rounding of the slab's sum
        sl = slice(k0, k0 + 16)
        for x, y in ((al, bh), (am, bm), (ah, bl), (am, bh), (ah, bm), (ah, bh)):
            acc = (acc.astype(np.float64) + np.einsum("rk,rk->r", x[:, sl].astype(np.float64), y[:, sl].astype(np.float64))).astype(np.float32)
    plain = np.zeros(rows, np.float32)
    for k0 in range(0, K, 2):        # the f32 MFMA: two products per instruction, one rounding each
        plain = (plain.astype(np.float64) + np.einsum("rk,rk->r", a[:, k0:k0 + 2].astype(np.float64),
                                                       b[:, k0:k0 + 2].astype(np.float64))).astype(np.float32)
    scale = np.abs(ref).max()
    e3, e1 = np.abs(acc - ref).max() / scale, np.abs(plain - ref).max() / scale
    assert e3 < 2e-6 and e3 <= 1.25 * e1 + 1e-8, (e3, e1)
