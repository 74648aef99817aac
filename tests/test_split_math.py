"""The arithmetic behind DSPN_MATH_F32_BF16X3 (dspnet_amd/csrc/conv.hip: split3 + six bf16 products), restated in numpy so
that its error bounds are checked where no GPU is needed: the three-piece representation and the six-term product."""
import numpy as np


def bf16_rne(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32: what v_cvt_pk_bf16_f32 does"""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    p0 = bf16_rne(x)
    r1 = (x - p0).astype(np.float32)
    p1 = bf16_rne(r1)
    r2 = (r1 - p1).astype(np.float32)
    p2 = bf16_rne(r2)
    return p0, p1, p2, r1, r2


def samples(n, seed):
    g = np.random.default_rng(seed)
    x = (g.standard_normal(n) * np.exp(g.uniform(-20, 20, n))).astype(np.float32)
    edge = np.array([0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 255.0, 256.0, 257.0, 0.1, 1e-30, -3e30, 2.0 ** -100,
                     np.float32(1.0 + 2.0 ** -8), np.float32(1.0 + 2.0 ** -8 + 2.0 ** -23), np.float32(1.0 + 3 * 2.0 ** -9)], np.float32)
    return np.concatenate([x, edge])


def test_the_two_residuals_are_exact_and_three_pieces_hold_x_to_half_an_ulp():
    x = samples(400000, 1)
    p0, p1, p2, r1, r2 = split3(x)
    x64 = x.astype(np.float64)
    assert np.array_equal(r1.astype(np.float64), x64 - p0)            # x - p0 is exact in float32
    assert np.array_equal(r2.astype(np.float64), x64 - p0 - p1)       # ... and so is the second residual
    err = np.abs(x64 - (p0.astype(np.float64) + p1 + p2))
    assert np.all(err <= 2.0 ** -24 * np.abs(x64))                   # half an fp32 ulp: as good as fp32 can hold x at all
    nz = x != 0
    assert np.all(np.abs(p1[nz]) <= 2.0 ** -8 * np.abs(x64[nz]) * (1 + 2.0 ** -7))
    assert np.all(np.abs(p2[nz]) <= 2.0 ** -16 * np.abs(x64[nz]) * (1 + 2.0 ** -6))


def test_six_partial_products_are_exact_in_float32_and_their_sum_is_an_fp32_product():
    x, w = samples(200000, 2), samples(200000, 3)[::-1].copy()
    px, pw = split3(x)[:3], split3(w)[:3]
    pairs = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]            # the kernel's order: small terms first
    total = np.zeros(x.shape, np.float64)
    for a, b in pairs:
        prod64 = px[a].astype(np.float64) * pw[b].astype(np.float64)
        prod32 = (px[a] * pw[b]).astype(np.float32)
        ok = np.isfinite(prod32) & ((np.abs(prod64) >= 2.0 ** -120) | (prod64 == 0))     # away from overflow / the denormal range
        assert np.array_equal(prod32[ok].astype(np.float64), prod64[ok])    # 8 x 8 significant bits: exact in the accumulator's format
        total += prod64
    exact = x.astype(np.float64) * w.astype(np.float64)
    ok = np.isfinite(exact) & (np.abs(exact) > 2.0 ** -100) & (np.abs(exact) < 2.0 ** 100)
    rel = np.abs(total[ok] - exact[ok]) / np.abs(exact[ok])
    assert rel.max() <= 2.0 ** -22                 # worst case: the three dropped pairs + the representation error
    assert np.sqrt(np.mean(rel ** 2)) <= 2.0 ** -25    # typical: below the rounding of ONE fp32 product (2^-24)
    # the plain bf16 mode of configs[3] for comparison: three orders of magnitude coarser
    rel_bf16 = np.abs(px[0].astype(np.float64)[ok] * pw[0].astype(np.float64)[ok] - exact[ok]) / np.abs(exact[ok])
    assert rel_bf16.max() > 2.0 ** -9 and np.sqrt(np.mean(rel_bf16 ** 2)) > 2.0 ** -10
