"""Pins the CPU oracle (oracle/) against the reference's own known-answer / property tests.

Each test names the Java test it re-expresses.  FT: = main/boofcv-feature/src/test/java/boofcv/,
IT: = main/boofcv-ip/src/test/java/boofcv/ under the reference tree.  No GPU, no product code here.
"""
import math

import numpy as np
import pytest


# ---------------------------------------------------------------------------------------------------
# java.util.Random: values fixed by the Java SE specification (widely published first outputs)
# ---------------------------------------------------------------------------------------------------
def test_java_random_known_values(orc):
    assert orc.JavaRandom(0).nextInt() == -1155484576
    assert orc.JavaRandom(42).nextInt() == -1170105035
    assert orc.JavaRandom(0).nextDouble() == 0.730967787376657
    assert orc.JavaRandom(0).nextFloat() == np.float32(0.73096776)
    assert abs(orc.JavaRandom(0).nextGaussian() - 0.8025330637390305) < 1e-15
    r = orc.JavaRandom(0)
    assert [r.nextInt(10) for _ in range(5)] == [0, 8, 9, 7, 5]
    # power-of-two bound: (bound * next(31)) >> 31 == the top log2(bound) bits of next(32)
    r = orc.JavaRandom(0); q = orc.JavaRandom(0)
    assert [r.nextInt(16) for _ in range(8)] == [(q.nextInt() & 0xFFFFFFFF) >> 28 for _ in range(8)]


# ---------------------------------------------------------------------------------------------------
# IT:alg/transform/ii/impl/TestImplIntegralImageOps.java
# ---------------------------------------------------------------------------------------------------
W, H = 20, 30


def test_integral_transform_and_subimage(orc):  # :69-89
    rand = orc.JavaRandom(234)
    img = rand.fillUniform(orc.Gray(W, H), 0, 100)
    for sub in (False, True):
        a = img.sub_image_of() if sub else img
        b = orc.Gray(W, H).sub_image_of(3, 2) if sub else orc.Gray(W, H)
        orc.lib().orc_integral(a.c(), b.c())
        expected = np.cumsum(np.cumsum(img.array().astype(np.float64), axis=0), axis=1)
        assert np.max(np.abs(expected - b.array())) < 1e-1


def test_block_sums_on_ones(orc):  # :131-145, :153-180
    ii = orc.integral(orc.Gray.from_array(np.ones((H, W), np.float32)))
    L = orc.lib()
    assert L.orc_block_unsafe(ii.c(), 4, 5, 8, 8) == 12
    assert L.orc_block_zero(ii.c(), 4, 5, 8, 8) == 12
    assert L.orc_block_zero(ii.c(), -1, -2, 2, 3) == 12
    assert L.orc_block_zero(ii.c(), W - 2, H - 3, W + 1, H + 3) == 2
    assert L.orc_block_zero(ii.c(), 3, -4, -1, -1) == 0
    assert L.orc_block_zero(ii.c(), W + 1, H + 2, W + 6, H + 8) == 0


def test_convolve_sparse_equals_dense(orc):  # :97-123
    import ctypes as C
    rand = orc.JavaRandom(234)
    ii = rand.fillUniform(orc.Gray(W, H), 0, 1000)
    blocks = np.array([-2, -2, 1, 1, -2, -1, 1, 0], dtype=np.int32)
    scales = np.array([1, 2], dtype=np.int32)
    A = ii.array().astype(np.float64)

    def bz(x0, y0, x1, y1):  # dense restatement of block_zero in float64
        x0, y0, x1, y1 = min(x0, W - 1), min(y0, H - 1), min(x1, W - 1), min(y1, H - 1)
        g = lambda x, y: A[y, x] if x >= 0 and y >= 0 else 0.0
        return g(x1, y1) - g(x1, y0) - g(x0, y1) + g(x0, y0)

    for (x, y) in [(0, 0), (10, 12), (19, 29)]:
        found = orc.lib().orc_convolve_sparse_blocks(ii.c(), 2, blocks.ctypes.data_as(C.POINTER(C.c_int)), scales.ctypes.data_as(C.POINTER(C.c_int)), x, y)
        exp = sum(bz(x + blocks[4 * i], y + blocks[4 * i + 1], x + blocks[4 * i + 2], y + blocks[4 * i + 3]) * scales[i] for i in range(2))
        assert abs(found - exp) < 1e-2 * max(1, abs(exp)) * 1e-2 + 1e-1  # fp32 sums of O(1e3) values


# ---------------------------------------------------------------------------------------------------
# FT:alg/feature/detect/intensity/TestIntegralImageFeatureIntensity.java:45-65 (+ impl test :44-69)
# ---------------------------------------------------------------------------------------------------
def test_hessian_inner_border_equals_naive(orc):
    rand = orc.JavaRandom(234)
    ii = orc.integral(rand.fillUniform(orc.Gray(60, 70), 0, 50))
    for skip in range(1, 5):
        expected = orc.hessian(ii, skip, 9, naive=True)
        found = orc.hessian(ii, skip, 9)
        assert expected.array().shape == (70 // skip, 60 // skip)
        assert np.max(np.abs(expected.array() - found.array())) <= 1e-4
    # the octave schedule's other sizes and a multi-threaded run agree too
    for skip, size in [(1, 15), (2, 27), (2, 51), (4, 27)]:
        e = orc.hessian(ii, skip, size, naive=True).array()
        f = orc.hessian(ii, skip, size, threads=4).array()
        assert np.max(np.abs(e - f)) <= 1e-4 * max(1.0, np.max(np.abs(e)))


# ---------------------------------------------------------------------------------------------------
# FT:alg/feature/detect/extract/GenericNonMaxTests.java (strict, max only) + GeneralNonMaxSuppressionChecks.java:231
# ---------------------------------------------------------------------------------------------------
NW, NH = 30, 40


def _blank():
    return np.zeros((NH, NW), np.float32)


def test_nonmax_border_maximum(orc):  # :94-124
    a = _blank(); a[1, 0] = 90; a[1, 1] = 30
    assert len(orc.nonmax(orc.Gray.from_array(a), 1, 5, 0)) == 1
    assert len(orc.nonmax(orc.Gray.from_array(a), 1, 5, 1)) == 0


def test_nonmax_along_image_border(orc):  # :126-150
    a = _blank()
    a[0, NW // 2] = 90; a[NH - 1, NW // 2] = 90; a[NH // 2, 0] = 90; a[NH // 2, NW - 1] = 90
    assert len(orc.nonmax(orc.Gray.from_array(a), 2, 5, 0)) == 4


def test_nonmax_strict_rule(orc):  # :159-175
    a = _blank()
    a[5, 3] = 30; a[7, 5] = 30; a[7, 7] = 30
    a[5, 2] = -30; a[7, 4] = -30; a[7, 6] = -30
    assert len(orc.nonmax(orc.Gray.from_array(a), 2, 5, 0)) == 0


def test_nonmax_ignores_max_value(orc):  # GeneralNonMaxSuppressionChecks.java:231
    a = _blank(); a[10, 10] = orc.MAX_VALUE_F32; a[20, 20] = 50
    found = orc.nonmax(orc.Gray.from_array(a), 2, 5, 0)
    assert found.tolist() == [[20, 20]]


def test_nonmax_block_equals_naive_as_sets(orc):  # :200-262
    rand = orc.JavaRandom(2134)
    for use_sub in (False, True):
        for radius in (1, 2, 3, 4):
            for _ in range(10):
                img = rand.fillGaussian(orc.Gray(NW, NH), 0, 3, -100, 100)
                if use_sub:
                    img = img.sub_image_of(5, 4)
                found = orc.nonmax(img, radius, 0.6, 0)
                naive = orc.nonmax(img, radius, 0.6, 0, naive=True)
                assert len(found) > 0 and len(found) == len(naive)
                assert set(map(tuple, found)) == set(map(tuple, naive))
                # block-raster order: keys strictly increasing
                step = radius + 1
                keys = [(y // step) * 1000 + (x // step) for x, y in found]
                assert keys == sorted(keys) and len(set(keys)) == len(keys)
                # threaded variant merges in the same order
                assert np.array_equal(orc.nonmax(img, radius, 0.6, 0, threads=3), found)


# ---------------------------------------------------------------------------------------------------
# FT:alg/feature/detect/interest/GenericFeatureDetectorTests.java:52-123 with TestFastHessianFeatureDetector.java:37-40
# ---------------------------------------------------------------------------------------------------
def _checkered(orc, w=80, h=90, sq=10, seed=234):
    """GenericFeatureDetectorTests.renderCheckered (:151-167): 50/0 squares + U[-5,5) noise."""
    yy, xx = np.mgrid[0:h, 0:w]
    a = np.where(((xx // sq) + (yy // sq)) % 2 == 0, 50.0, 0.0).astype(np.float32)
    noise = orc.JavaRandom(seed).fillUniform(orc.Gray(w, h), -5, 5).array()
    return orc.Gray.from_array(a + noise)


def test_fast_hessian_detector_generic(orc):
    # FH(extractor(r=1,thr=1,border 5), maxPerScale, 1, 9, 4, 4, 6)
    img = _checkered(orc)
    ii = orc.integral(img)
    n_all = len(orc.fh_detect(ii, orc.FhCfg(1.0, 1, -1, 1, 9, 4, 4, 6)))
    n_20 = len(orc.fh_detect(ii, orc.FhCfg(1.0, 1, 20, 1, 9, 4, 4, 6)))
    n_10 = len(orc.fh_detect(ii, orc.FhCfg(1.0, 1, 10, 1, 9, 4, 4, 6)))
    assert n_all > 0 and n_10 <= n_20 <= n_all and n_10 < n_all  # checkFlags / maxFeatures monotone
    # idempotent (checkMultipleCalls)
    assert np.array_equal(orc.fh_detect(ii, orc.FhCfg(1.0, 1, -1, 1, 9, 4, 4, 6)), orc.fh_detect(ii, orc.FhCfg(1.0, 1, -1, 1, 9, 4, 4, 6)))
    # blank image has fewer features than the checkered one
    blank = orc.integral(orc.Gray.from_array(np.full((90, 80), 50, np.float32)))
    assert len(orc.fh_detect(blank, orc.FhCfg(1.0, 1, -1, 1, 9, 4, 4, 6))) < n_all


def test_fast_hessian_threads_do_not_change_order(orc):
    ii = orc.integral(orc.noise_image(200, 150, 77))
    a = orc.fh_detect(ii, threads=1)
    b = orc.fh_detect(ii, threads=4)
    assert len(a) > 50 and np.array_equal(a, b)


# ---------------------------------------------------------------------------------------------------
# FT:alg/feature/orientation/GenericOrientationIntegralTests.java:98-170,200-216
# ---------------------------------------------------------------------------------------------------
def _oriented(orc, angle, w=30, h=40):
    yy, xx = np.mgrid[0:h, 0:w]
    a = (10 * (xx * math.cos(angle) + yy * math.sin(angle))).astype(np.float32)
    return orc.integral(orc.Gray.from_array(a))


def _adist(a, b):
    d = abs(a - b) % (2 * math.pi)
    return min(d, 2 * math.pi - d)


@pytest.mark.parametrize("kind,tol,cfg", [
    ("sliding", math.pi / 9, lambda o: o.OriCfg(0.3, 1.0, math.pi / 3, 3, 0.0, 4)),   # TestImplOrientationSlidingWindowIntegral r=3
    ("sliding", math.pi / 9, lambda o: o.OriCfg(0.3, 1.0, math.pi / 3, 3, -1.0, 4)),
    ("average", 0.01, lambda o: o.OriCfg(0.5, 1.0, 0.0, 4, 0.0, 2)),                   # TestImplOrientationAverageGradientIntegral r=4
    ("average", 0.01, lambda o: o.OriCfg(0.5, 1.0, 0.0, 4, -1.0, 2)),
])
def test_orientation_planar_ramp(orc, kind, tol, cfg):
    c = cfg(orc)
    N = 2 * int(math.pi / tol)
    for i in range(N):
        angle = i * tol
        ii = _oriented(orc, angle)
        found = orc.orientation(ii, 15, 20, 10, kind, c)
        assert _adist(angle, found) < tol
    angle = (N // 2) * tol
    ii = _oriented(orc, angle)
    for radius in (10, 15, 7.5):  # setScale
        assert _adist(angle, orc.orientation(ii, 15, 20, radius, kind, c)) < tol
    # checkBorderExplode: no crash along the border
    for y in range(40):
        orc.orientation(ii, 0, y, 10, kind, c); orc.orientation(ii, 29, y, 10, kind, c)


def test_orientation_defaults_planar_ramp(orc):
    for angle in (0.0, 0.5, 2.0, -1.2, 3.0):
        ii = _oriented(orc, angle, 120, 100)
        assert _adist(angle, orc.orientation(ii, 60, 50, 2 * 2.0, "sliding")) < math.pi / 9
        assert _adist(angle, orc.orientation(ii, 60, 50, 2 * 2.0, "average")) < 0.01


# ---------------------------------------------------------------------------------------------------
# FT:alg/feature/describe/BaseTestDescribeSurf.java:139-212 (both DescribePointSurf and DescribePointSurfMod)
# ---------------------------------------------------------------------------------------------------
def _xramp(orc, w=50, h=60):
    yy, xx = np.mgrid[0:h, 0:w]
    return orc.integral(orc.Gray.from_array((1.0 * xx + 0.0 * yy).astype(np.float32)))


@pytest.mark.parametrize("stable", [True, False])
def test_surf_descriptor_analytic(orc, stable):
    ii = orc.integral(orc.Gray.from_array(np.full((60, 50), 50, np.float32)))
    f, _ = orc.describe(ii, 20, 20, 0.75, 1, stable)
    assert np.all(np.abs(f) < 1e-4)  # features_constant

    ii = _xramp(orc)
    f, _ = orc.describe(ii, 15, 15, 0.0, 1, stable)  # features_increasing, along x
    for i in range(0, 64, 4):
        assert abs(f[i] - f[i + 1]) < 1e-4 and f[i] > 0 and abs(f[i + 2]) < 1e-4 and abs(f[i + 3]) < 1e-4
    f, _ = orc.describe(ii, 15, 15, math.pi / 2, 1, stable)  # along y
    for i in range(0, 64, 4):
        assert abs(-f[i + 2] - f[i + 3]) < 1e-4 and f[i + 2] < 0 and abs(f[i]) < 1e-4 and abs(f[i + 1]) < 1e-4
    f, _ = orc.describe(ii, 25, 25, 0.0, 1.5, stable)  # features_fraction
    for i in range(0, 64, 4):
        assert abs(f[i] - f[i + 1]) < 1e-4 and f[i] > 0 and abs(f[i + 2]) < 1e-4 and abs(f[i + 3]) < 1e-4
    assert abs(np.linalg.norm(f) - 1) < 1e-12


@pytest.mark.parametrize("stable", [True, False])
def test_surf_descriptor_subimage_scale_rotation_border(orc, stable):
    rand = orc.JavaRandom(234)
    ii = rand.fillUniform(orc.Gray(50, 60), 0, 100)  # the reference fills the integral image directly
    a, wa = orc.describe(ii, 25, 30, 0, 1, stable)
    b, wb = orc.describe(ii.sub_image_of(), 25, 30, 0, 1, stable)
    assert wa == wb and np.array_equal(a, b)  # checkSubImage
    c, _ = orc.describe(ii, 25, 30, 0, 1.5, stable)
    d, _ = orc.describe(ii, 25, 30, 1, 1, stable)
    assert np.max(np.abs(a - c)) > 1e-4 and np.max(np.abs(a - d)) > 1e-4  # changeScale / changeRotation
    for i in range(10):  # checkBorder: must not blow up
        ang = 2 * math.pi * i / 10
        f0, _ = orc.describe(ii, 0, 0, ang, 1, stable)
        f1, _ = orc.describe(ii, 49, 59, ang, 1, stable)
        assert np.all(np.isfinite(f0)) and np.all(np.isfinite(f1))


# ---------------------------------------------------------------------------------------------------
# FT:abst/feature/detdesc/GenericTestsDetectDescribePoint.java:83-205 ; TestWrapDetectDescribeSurf_MT.java:54-103
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("stable", [True, False])
def test_detect_describe_generic(orc, stable):
    rand = orc.JavaRandom(234)
    s = orc.Surf(stable)
    for _ in range(10):
        img = rand.fillUniform(orc.Gray(100, 120), 0, 100)
        n = s.detect(img)
        assert n > 5
        xys, ang, white, desc = s.fetch()
        assert not np.any(np.isnan(desc)) and not np.any(np.isnan(ang))
        assert np.allclose(np.linalg.norm(desc, axis=1), 1, atol=1e-12)
    # sub-image == full, repeat call == first call
    n2 = s.detect(img.sub_image_of())
    r2 = s.fetch()
    assert n2 == n and all(np.array_equal(p, q) for p, q in zip((xys, ang, white, desc), r2))
    s.detect(img)
    assert all(np.array_equal(p, q) for p, q in zip((xys, ang, white, desc), s.fetch()))


def test_detect_describe_mt_equals_st(orc):
    img = orc.JavaRandom(234).fillUniform(orc.Gray(400, 300), 0, 100)
    s = orc.Surf(True)
    n1 = s.detect(img, threads=1); r1 = s.fetch()
    n4 = s.detect(img, threads=4); r4 = s.fetch()
    assert n1 == n4 and n1 > 200
    assert all(np.array_equal(p, q) for p, q in zip(r1, r4))  # deterministic merge order in the oracle


# ---------------------------------------------------------------------------------------------------
# FT:alg/descriptor/TestDescriptorDistance.java:47-56,166-188
# ---------------------------------------------------------------------------------------------------
def test_descriptor_distance(orc):
    import ctypes as C
    a = np.array([1, 2, 3, 4, 5], np.float64); b = np.array([2, -1, 7, -8, 10], np.float64)
    P = lambda x: x.ctypes.data_as(C.POINTER(C.c_double))
    assert orc.lib().orc_euclidean_sq(P(a), P(b), 5) == 195
    hw = orc.lib().orc_hamming_word
    assert [hw(0), hw(0x0800), hw(0x0001), hw(0x0101), hw(0x000F)] == [0, 1, 1, 2, 4]
    assert hw(np.int32(np.uint32(0xF000000F)).item()) == 8
    rand = orc.JavaRandom(234)
    for _ in range(20):
        x = np.array([rand.nextInt() for _ in range(16)], np.int32); y = np.array([rand.nextInt() for _ in range(16)], np.int32)
        exp = sum(bin((int(p) ^ int(q)) & 0xFFFFFFFF).count("1") for p, q in zip(x, y))
        I = lambda v: v.ctypes.data_as(C.POINTER(C.c_int32))
        assert orc.lib().orc_hamming(I(x), I(y), 16) == exp


# ---------------------------------------------------------------------------------------------------
# FT:alg/feature/associate/TestAssociateGreedy.java:38-105 (score = ScoreAssociateEuclidean_F64)
# ---------------------------------------------------------------------------------------------------
def _col(*v):
    return np.array(v, np.float64).reshape(-1, 1)


def test_greedy_basic(orc):
    pairs, fit = orc.associate_l2(_col(1, 2, 3, 4), _col(3, 4, 1, 40), 0.5, False, sqrt_score=True)
    assert pairs.tolist() == [2, -1, 0, 1]
    assert fit[0] == 0 and fit[2] == 0 and fit[3] == 0


def test_greedy_max_error(orc):
    assert orc.associate_l2(_col(1, 2, 3, 4), _col(3, 4, 1.1, 40), 10, False, sqrt_score=True)[0][1] == 2
    assert orc.associate_l2(_col(1, 2, 3, 4), _col(3, 4, 1.1, 40), 0.1, False, sqrt_score=True)[0][1] == -1


def test_greedy_backwards(orc):
    pairs, fit = orc.associate_l2(_col(1, 2, 3, 8), _col(3, 4, 1, 10), 10, True, sqrt_score=True)
    assert pairs.tolist() == [2, -1, 0, 3]
    assert fit[0] == 0 and fit[2] == 0 and fit[3] == 2
    assert fit[1] == orc.MAX_VALUE_F64  # rejected by backwards validation
    # squared score variant used on the product path
    pairs, fit = orc.associate_l2(_col(1, 2, 3, 8), _col(3, 4, 1, 10), 10, True)
    assert pairs.tolist() == [2, -1, 0, 3] and fit[3] == 4


def test_greedy_threads_equal(orc):
    rng = np.random.default_rng(5)
    a = rng.normal(size=(200, 8)); b = rng.normal(size=(180, 8))
    p1, f1 = orc.associate_l2(a, b, threads=1); p4, f4 = orc.associate_l2(a, b, threads=4)
    assert np.array_equal(p1, p4) and np.array_equal(f1, f4)


# ---------------------------------------------------------------------------------------------------
# FT:alg/feature/describe/BaseTestDescribePointBinaryCompare.java:141-191
# ---------------------------------------------------------------------------------------------------
def test_brief_bit_order_and_border(orc):
    rand = orc.JavaRandom(234)
    img = rand.fillUniform(orc.Gray(40, 50), 0, 100)
    sp, cp = orc.brief_definition(123, 5, 20)  # 20 pairs -> one word
    assert np.all(np.hypot(sp[:, 0], sp[:, 1]) < 5) and np.all(cp[:, 0] == np.arange(20)) and np.all((cp[:, 1] >= 0) & (cp[:, 1] < 20))
    A = img.array()
    cx, cy = 20, 25
    d = orc.brief_describe(img, [[cx, cy]], 5, sp, cp)[0]
    n = len(cp)
    for i in range(n):
        a = A[cy + sp[cp[i, 0], 1], cx + sp[cp[i, 0], 0]]; b = A[cy + sp[cp[i, 1], 1], cx + sp[cp[i, 1], 0]]
        bit = (int(d[0]) >> (n - i - 1)) & 1  # bit `compare.length-i-1` <=> pair i
        assert bit == (1 if a < b else 0)
    # 512-bit default definition: pair i of a 32-group at bit 31-(i mod 32)
    sp, cp = orc.brief_definition()
    img = rand.fillUniform(orc.Gray(80, 70), 0, 100); A = img.array()
    d = orc.brief_describe(img, [[40.7, 35.2]], 16, sp, cp)[0]
    for i in range(512):
        a = A[35 + sp[cp[i, 0], 1], 40 + sp[cp[i, 0], 0]]; b = A[35 + sp[cp[i, 1], 1], 40 + sp[cp[i, 1], 0]]
        assert ((int(d[i // 32]) >> (31 - i % 32)) & 1) == (1 if a < b else 0)
    # border path: pairs with a sample outside are skipped without shifting; never reads outside
    d = orc.brief_describe(img, [[2, 3], [79, 69], [0, 0]], 16, sp, cp)
    assert d.shape == (3, 16)


# ---------------------------------------------------------------------------------------------------
# IT:alg/filter/convolve/*, IT:alg/filter/blur/TestBlurImageOps.java, IT:alg/filter/derivative/TestGradientSobel.java
# ---------------------------------------------------------------------------------------------------
def test_convolution_equivalences(orc):
    rand = orc.JavaRandom(234)
    img = rand.fillUniform(orc.Gray(35, 28), 0, 50)
    A = img.array().astype(np.float64)
    for r in (1, 2, 3, 5, 6):
        k = orc.gaussian1d_f32(-1, r)
        assert len(k) == 2 * r + 1 and abs(float(np.sum(k)) - 1) < 1e-6
        h = orc.conv("h", k, r, img).array()
        v = orc.conv("v", k, r, img).array()
        eh = np.zeros_like(A); ev = np.zeros_like(A)
        for i in range(2 * r + 1):
            eh[:, r:35 - r] += A[:, i:35 - 2 * r + i] * float(k[i])
            ev[r:28 - r, :] += A[i:28 - 2 * r + i, :] * float(k[i])
        assert np.max(np.abs(h - eh)) < 1e-3 and np.max(np.abs(v - ev)) < 1e-3
        # normalised border: interior equals no-border result; border = renormalised partial sums
        nh = orc.conv("norm_h", k, r, img).array()
        assert np.array_equal(nh[:, r:35 - r], h[:, r:35 - r])
        x = 0
        part = sum(A[3, j] * float(k[j - x + r]) for j in range(0, r + 1)) / sum(float(k[j - x + r]) for j in range(0, r + 1))
        assert abs(nh[3, 0] - part) < 1e-3
        # Gaussian blur = normalised H then V; constant image stays constant
    flat = orc.Gray.from_array(np.full((28, 35), 7, np.float32))
    assert np.max(np.abs(orc.gaussian_blur(flat, -1, 2).array() - 7)) < 1e-5
    b = orc.gaussian_blur(img, -1, 2)
    k = orc.gaussian1d_f32(-1, 2)
    assert np.array_equal(b.array(), orc.conv("norm_v", k, 2, orc.conv("norm_h", k, 2, img)).array())


def test_gradients_equal_kernel_convolution(orc):
    rand = orc.JavaRandom(234)
    img = rand.fillUniform(orc.Gray(31, 26), 0, 50)
    A = img.array().astype(np.float64)
    dx, dy = orc.gradient("sobel", img)
    kx = np.array([[-0.25, 0, 0.25], [-0.5, 0, 0.5], [-0.25, 0, 0.25]]); ky = kx.T
    ex = np.zeros_like(A); ey = np.zeros_like(A)
    for i in range(3):
        for j in range(3):
            ex[1:-1, 1:-1] += A[i:26 - 2 + i, j:31 - 2 + j] * kx[i, j]
            ey[1:-1, 1:-1] += A[i:26 - 2 + i, j:31 - 2 + j] * ky[i, j]
    assert np.max(np.abs(dx.array() - ex)) < 1e-4 and np.max(np.abs(dy.array() - ey)) < 1e-4
    tx, ty = orc.gradient("three", img)
    assert np.allclose(tx.array()[1:-1, 1:-1], (A[1:-1, 2:] - A[1:-1, :-2]) * 0.5, atol=1e-5)
    assert np.allclose(ty.array()[1:-1, 1:-1], (A[2:, 1:-1] - A[:-2, 1:-1]) * 0.5, atol=1e-5)
    # zero-value border policy fills the frame
    bx, by = orc.gradient("sobel", img, border_zero=True)
    Ap = np.pad(A, 1)
    e = sum(Ap[0 + i, 0 + j] * kx[i, j] for i in range(3) for j in range(3))
    assert abs(bx.array()[0, 0] - e) < 1e-4
    assert np.array_equal(bx.array()[1:-1, 1:-1], dx.array()[1:-1, 1:-1])


# ---------------------------------------------------------------------------------------------------
# Gaussian tables used by orientation / SURF (I:factory/filter/kernel/FactoryKernelGaussian.java) -- "sums to one" property
# ---------------------------------------------------------------------------------------------------
def test_gaussian_tables(orc):
    w = orc.gaussian2d_f64(-1, 8)
    assert w.shape == (17, 17) and abs(w.sum() - 1) < 1e-12 and np.allclose(w, w.T) and w[8, 8] == w.max()
    g = orc.gaussian_width(2.5, 4)
    assert g.shape == (4, 4) and abs(g.sum() - 1) < 1e-12 and np.allclose(g, g[::-1, ::-1])
    g9 = orc.gaussian_width(2.5, 9)
    assert g9.shape == (9, 9) and abs(g9.sum() - 1) < 1e-12 and g9[4, 4] == g9.max()
    g20 = orc.gaussian_width(4.5, 20)
    assert g20.shape == (20, 20) and abs(g20.sum() - 1) < 1e-12


# ---------------------------------------------------------------------------------------------------
# Down-sampling convolution + discrete pyramid
# ---------------------------------------------------------------------------------------------------
def test_down_convolve_util_known_answers(orc):  # IT:alg/filter/convolve/down/TestUtilDownConvolve.java:31-55
    L = orc.lib()
    for expect, args in [(8, (10, 1, 1)), (7, (10, 1, 2)), (8, (10, 2, 1)), (6, (10, 2, 2)), (6, (10, 2, 3)), (4, (10, 2, 4)),
                         (6, (10, 3, 1)), (6, (10, 3, 2)), (6, (10, 3, 3)), (3, (10, 3, 4)), (4, (11, 4, 2))]:
        assert L.orc_down_max_side(*args) == expect
    for expect, args in [(1, (1, 1)), (2, (1, 2)), (3, (1, 3)), (2, (2, 1)), (2, (2, 2)), (4, (2, 3))]:
        assert L.orc_down_offset(*args) == expect


def _down_naive64(a, ker, skip, vertical):
    """Independent restatement: normalised clipped window at every skip-th pixel, in float64."""
    if vertical:
        return _down_naive64(a.T, ker, skip, False).T
    h, w = a.shape
    r = len(ker) // 2
    out = np.zeros((h, w // skip))
    for X in range(w // skip):
        x = X * skip
        lo, hi = max(0, x - r), min(w - 1, x + r)
        kk = np.asarray(ker[lo - x + r:hi - x + r + 1], dtype=np.float64)
        out[:, X] = (a[:, lo:hi + 1].astype(np.float64) @ kk) / kk.sum()
    return out


def test_down_convolve_normalized_vs_naive(orc):  # IT:alg/filter/convolve/TestConvolveImageDownNormalized.java:30-50
    rand = orc.JavaRandom(0xFF)
    for i in range(2):
        w, h = 15 + i, 20 + i
        for radius in (1, 2, 3, 10):  # 10: kernel wider than the image -> naive path
            ker = orc.gaussian1d_f32(-1, radius)
            src = rand.fillUniform(orc.Gray(w, h), 1, 10)
            for kind in ("h", "v"):
                for sub in (False, True):
                    s = src.sub_image_of() if sub else src
                    got = orc.conv_down(kind, ker, s, 2).array()
                    want = _down_naive64(src.array(), ker, 2, kind == "v")
                    # the Gaussian kernel sums to 1 within float rounding, so interior (plain sum) and naive
                    # (sum / kernel sum) agree to the reference test's tolerance
                    assert np.abs(got - want).max() < 1e-4 * 10


def test_down_convolve_interior_is_plain_sum_border_is_normalised(orc):
    """The un-normalised interior is visible with a kernel that does not sum to one."""
    rand = orc.JavaRandom(3)
    src = rand.fillUniform(orc.Gray(30, 17), 0, 50)
    ker = np.array([1, 2, 3, 2, 1], dtype=np.float32)
    got = orc.conv_down("h", ker, src, 2).array()
    a = src.array().astype(np.float64)
    full = np.array([1, 2, 3, 2, 1.0])
    # left edge x = 0: window clipped to taps 0..+2, divided by their weight
    assert np.allclose(got[:, 0], (a[:, 0] * 3 + a[:, 1] * 2 + a[:, 2]) / 6, rtol=1e-6)
    # interior x = 10 and the last interior x = computeMaxSide(30,2,2) = 26: plain sum, NOT divided by 9
    assert np.allclose(got[:, 5], a[:, 8:13] @ full, rtol=1e-6)
    assert np.allclose(got[:, 13], a[:, 24:29] @ full, rtol=1e-6)
    # right edge x = 28: taps -2..+1 only
    assert np.allclose(got[:, 14], (a[:, 26:30] @ full[:4]) / 8, rtol=1e-6)


def test_down_convolve_rejects_bad_shapes(orc):
    src = orc.Gray(20, 20)
    ker = orc.gaussian1d_f32(-1, 2)
    with pytest.raises(ValueError):
        orc.conv_down("h", ker, src, 0, out=orc.Gray(20, 20))
    with pytest.raises(ValueError):
        orc.conv_down("h", ker, src, 2, out=orc.Gray(9, 20))
    with pytest.raises(ValueError):
        orc.conv_down("v", ker, src, 2, out=orc.Gray(20, 9))


def test_pyramid_discrete_sample_blur_update(orc):  # IT:alg/transform/pyramid/TestPyramidDiscreteSampleBlur.java:44-92
    width, height = 80, 120
    rand = orc.JavaRandom(234)
    inp = rand.fillUniform(orc.Gray(width, height), 0, 100)
    ker = orc.gaussian1d_f32(-1, 3)
    layers, _ = orc.pyramid(ker, 3, [1, 2, 4], inp)
    assert [l.shape for l in layers] == [(120, 80), (60, 40), (30, 20)]
    assert np.array_equal(layers[0], inp.array())
    conv = orc.conv("norm_v", ker, 3, orc.conv("norm_h", ker, 3, inp)).array()
    assert np.abs(conv[::2, ::2] - layers[1]).max() < 1e-4
    l1 = orc.Gray.from_array(layers[1])
    conv2 = orc.conv("norm_v", ker, 3, orc.conv("norm_h", ker, 3, l1)).array()
    assert np.abs(conv2[::2, ::2] - layers[2]).max() < 1e-4
    # checkModifiesLayersOnUpdate (GenericPyramidTests.java:52-64)
    assert all(l.sum() > 0 for l in layers)


def test_pyramid_sigmas_and_odd_sizes(orc):  # TestPyramidDiscreteSampleBlur.java:97-111
    ker = orc.gaussian1d_f32(-1, 3)
    inp = orc.JavaRandom(1).fillUniform(orc.Gray(41, 27), 0, 100)
    layers, sig = orc.pyramid(ker, 3, [1, 2, 4], inp)
    assert sig[0] == 0 and abs(sig[1] - 3) < 1e-8 and abs(sig[2] - 6.7082) < 1e-3
    # ImagePyramidBase.initialize: ceil sizes; the convolution fills floor(prev/skip), the rest stays 0
    assert [l.shape for l in layers] == [(27, 41), (14, 21), (7, 11)]
    assert np.all(layers[1][13, :] == 0) and np.all(layers[1][:, 20] == 0) and layers[1][:13, :20].min() > 0
    layers, sig = orc.pyramid(ker, 3, [2, 4, 8], inp)
    assert sig[0] == 0 and abs(sig[1] - 6) < 1e-8
    assert [l.shape for l in layers] == [(14, 21), (7, 11), (4, 6)]


# ---------------------------------------------------------------------------------------------------
# AssociateSurfBasic   FT:alg/feature/associate/TestAssociateSurfBasic.java
# ---------------------------------------------------------------------------------------------------
def _surf_desc(vals):
    d = np.zeros((len(vals), 64))
    d[:, 0] = vals
    return d


def test_associate_surf_basic_literals(orc):
    # checkAssociateByIntensity :44-62: different Laplacian signs are never associated, even when the other sign fits better
    m, un = orc.associate_surf_basic(_surf_desc([10]), [True], _surf_desc([0, 10]), [True, False], maxErr=20, backwards=True, sqrt_score=True)
    assert len(m) == 1 and m[0][1] == 0 and un == []
    # basicAssociation :64-109
    m, un = orc.associate_surf_basic(_surf_desc([10, 12, 5, 2344, 1000]), [True, True, False, False, False],
                                     _surf_desc([0, 10.1, 13, 0.1, 7]), [True, True, True, False, False], maxErr=20, backwards=True, sqrt_score=True)
    assert [(a, b) for a, b, _ in m] == [(0, 1), (1, 2), (2, 4)] and all(f != 0 for _, _, f in m)
    assert len(un) == 2 and not set(un) & {a for a, _, _ in m}
    # handleEmptyLists :153-: empty source or destination gives no matches
    assert orc.associate_surf_basic(_surf_desc([]), [], _surf_desc([10]), [True]) == ([], [])
    assert orc.associate_surf_basic(_surf_desc([10]), [True], _surf_desc([]), []) == ([], [])


# ---------------------------------------------------------------------------------------------------
# Colour SURF   FT:alg/feature/describe/TestDescribePointSurfPlanar.java:48-89
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("stable", [True, False])
def test_surf_planar_compare_to_single_band(orc, stable):
    rand = orc.JavaRandom(234)
    bands = [rand.fillUniform(orc.Gray(150, 120), 0, 200) for _ in range(3)]
    s = orc.Surf(stable)
    n = s.detect_planar(bands)
    assert n > 20
    pts, ang, white, desc = s.fetch()
    assert desc.shape == (n, 192) and np.allclose(np.linalg.norm(desc, axis=1), 1, atol=1e-12)
    # detector ran on the band average (ImplConvertPlanarToGray.average: (b0 + b1 + b2) / 3 in float)
    avg = ((bands[0].array() + bands[1].array()) + bands[2].array()) / np.float32(3)
    g = orc.Surf(stable)
    g.detect(orc.Gray.from_array(avg))
    gp, gang, gwhite, _ = g.fetch()
    assert np.array_equal(gp, pts) and np.array_equal(gwhite, white)
    # every band's block, renormalised on its own, equals the single-band descriptor of that band at the same (x, y, angle, scale)
    for b in range(3):
        bii = orc.integral(bands[b])
        for i in range(0, n, max(1, n // 25)):
            d1, _ = orc.describe(bii, pts[i, 0], pts[i, 1], ang[i], pts[i, 2], stable=stable)
            blk = desc[i, 64 * b:64 * (b + 1)]
            assert np.abs(blk / np.linalg.norm(blk) - d1).max() < 1e-8


# ---------------------------------------------------------------------------------------------------
# Gradient corner intensity
# ---------------------------------------------------------------------------------------------------
def test_ssd_corner_compare_to_manual(orc):  # FT:alg/feature/detect/intensity/impl/TestImplSsdCorner_F32.java:55-86
    width, height, radius = 40, 50, 4
    inp = orc.JavaRandom(234).fillUniform(orc.Gray(width, height), 0, 100)
    dx, dy = orc.gradient("sobel", inp, border_zero=True)   # the test uses an extended border; any gradient exercises the sums
    got = orc.corner_intensity(dx, dy, radius, "mocksum")
    x, y = dx.array().astype(np.float64), dy.array().astype(np.float64)
    prod = x * x + x * y + y * y
    for yy in range(radius, height - radius):
        for xx in range(radius, width - radius):
            assert abs(prod[yy - radius:yy + radius + 1, xx - radius:xx + radius + 1].sum() - got[yy, xx]) < 1   # the reference test's absolute tolerance
    assert np.all(got[:radius] == 0) and np.all(got[:, :radius] == 0) and np.all(got[-radius:] == 0) and np.all(got[:, -radius:] == 0)


def test_corner_scores_known_values(orc):  # TestShiTomasiCorner_F32 / TestHarrisCorner_F32: closed forms on a constant-gradient patch
    w = h = 12
    dx = orc.Gray.from_array(np.full((h, w), 2.0, np.float32)); dy = orc.Gray.from_array(np.full((h, w), 0.0, np.float32))
    st = orc.corner_intensity(dx, dy, 1, "shitomasi"); ha = orc.corner_intensity(dx, dy, 1, "harris", 0.04)
    # A = [[36, 0], [0, 0]]: smallest eigenvalue 0; Harris = det - k trace^2 = -0.04 * 36^2
    assert st[5, 5] == 0 and abs(ha[5, 5] + 0.04 * 36 * 36) < 1e-3
    dy = orc.Gray.from_array(np.full((h, w), 3.0, np.float32))
    st = orc.corner_intensity(dx, dy, 1, "shitomasi")
    # A = 9 * [[4, 6], [6, 9]]: eigenvalues 0 and 117
    assert abs(st[5, 5]) < 1e-3


# ---------------------------------------------------------------------------------------------------
# Remaining BOverride hooks: 2-D convolution, mean blur, median blur   IT:alg/filter/blur/TestBlurImageOps.java:60-170
# ---------------------------------------------------------------------------------------------------
def test_blur_mean_median_and_conv2d(orc):
    rand = orc.JavaRandom(234)
    img = rand.fillUniform(orc.Gray(25, 20), 0, 20)   # TestBlurImageOps: 20x25-ish, U[0,20)
    a = img.array().astype(np.float64)
    h, w = a.shape
    for radius in range(1, 5):
        # mean == normalised table convolution (BlurImageOps test: tolerance 2; the float running sums are far inside it)
        exp = np.zeros_like(a)
        for y in range(h):
            for x in range(w):
                exp[y, x] = a[max(0, y - radius):y + radius + 1, max(0, x - radius):x + radius + 1].mean()
        assert np.abs(orc.blur_mean(img, radius).array() - exp).max() < 1e-4
        # median == the (count/2)-th order statistic of the clipped window
        med = orc.blur_median(img, radius).array()
        for y in range(h):
            for x in range(w):
                win = np.sort(img.array()[max(0, y - radius):y + radius + 1, max(0, x - radius):x + radius + 1].ravel())
                assert med[y, x] == win[len(win) // 2]
    with pytest.raises(ValueError):
        orc.blur_mean(img, 0)
    with pytest.raises(ValueError):
        orc.blur_median(img, 0)
    # kernel wider than the image falls back to the normalised convolution
    small = rand.fillUniform(orc.Gray(4, 3), 0, 20)
    assert np.allclose(orc.blur_mean(small, 3).array(), small.array().mean(), rtol=1e-5)
    # 2-D convolution: unrolled (3..11) and standard forms agree to float rounding; frame untouched
    for kw in (3, 5, 4):
        k = np.arange(1, kw * kw + 1, dtype=np.float32).reshape(kw, kw) / (kw * kw)
        out = orc.Gray(25, 20); out.buf[:] = -7
        got = orc.conv2d(k, kw // 2, img, out).array()
        oL, oR = kw // 2, kw - kw // 2 - 1
        for y in range(oL, h - oR):
            for x in range(oL, w - oR):
                assert abs(got[y, x] - (a[y - oL:y - oL + kw, x - oL:x - oL + kw] * k).sum()) < 1e-3
        assert got[0, 0] == -7 and got[-1, -1] == -7


# ---------------------------------------------------------------------------------------------------
# Integer image variants, stage level
# ---------------------------------------------------------------------------------------------------
def test_integer_variants_stage_level(orc):
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, (70, 60), dtype=np.uint8)
    # IT:alg/transform/ii/impl/TestImplIntegralImageOps.java (GrayU8 -> GrayS32): ii(x, y) = sum of the pixels above and to the left, exactly
    ii = orc.integral_u8(img)
    assert np.array_equal(ii, img.astype(np.int64).cumsum(0).cumsum(1).astype(np.int32))
    # Hessian on the S32 integral: all-int box sums -> identical to a brute-force evaluation of computeHessian in exact arithmetic
    # (FT:alg/feature/detect/intensity/impl/TestImplIntegralImageFeatureIntensity.java:44-69 compares inner + border with naive, 1e-4)
    def block(x0, y0, x1, y1):
        h, w = ii.shape
        x0, y0, x1, y1 = min(x0, w - 1), min(y0, h - 1), min(x1, w - 1), min(y1, h - 1)
        g = lambda x, y: int(ii[y, x]) if x >= 0 and y >= 0 else 0
        return g(x1, y1) - g(x1, y0) - g(x0, y1) + g(x0, y0)
    for skip, size in [(1, 9), (2, 15), (3, 9)]:
        got = orc.hessian_s32(ii, skip, size)
        bw = size // 3; bh = size - bw - 1; r1 = bw // 2; r2 = bw + r1; r3 = bh // 2
        norm = np.float32(1.0) / np.float32(size * size)
        for (x, y) in [(0, 0), (5, 7), (got.shape[1] - 1, got.shape[0] - 1), (got.shape[1] // 2, got.shape[0] // 2), (1, got.shape[0] // 2)]:
            xx, yy = x * skip, y * skip
            dxx = block(xx - r2 - 1, yy - r3 - 1, xx + r2, yy + r3) - 3 * block(xx - r1 - 1, yy - r3 - 1, xx + r1, yy + r3)
            dyy = block(xx - r3 - 1, yy - r2 - 1, xx + r3, yy + r2) - 3 * block(xx - r3 - 1, yy - r1 - 1, xx + r3, yy + r1)
            dxy = (block(xx - bw - 1, yy - bw - 1, xx - 1, yy - 1) - block(xx, yy - bw - 1, xx + bw, yy - 1) + block(xx, yy, xx + bw, yy + bw)
                   - block(xx - bw - 1, yy, xx - 1, yy + bw))
            Dxx, Dyy, Dxy = np.float32(dxx) * norm, np.float32(dyy) * norm, np.float32(dxy) * norm
            exp = np.float32(np.float32(Dxx * Dyy) - np.float32(np.float32(np.float32(0.81) * Dxy) * Dxy))
            assert got[y, x] == exp, (skip, size, x, y)
    # BRIEF on GrayU8 (FT:alg/feature/describe/BaseTestDescribePointBinaryCompare.java:141-191): bit order, and border == inside when inside
    sp, cp = orc.brief_definition()
    inside = orc.brief_describe_u8(img, [[30, 35]], 16, sp, cp)[0]
    vals = lambda k: (int(img[35 + sp[cp[k, 0], 1], 30 + sp[cp[k, 0], 0]]), int(img[35 + sp[cp[k, 1], 1], 30 + sp[cp[k, 1], 0]]))
    for k in (0, 1, 31, 32, 500, 511):
        a, b = vals(k)
        assert ((int(inside[k // 32]) >> (31 - k % 32)) & 1) == (1 if a < b else 0)
    big = np.zeros((100, 100), np.uint8); big[10:80, 20:80] = img
    assert np.array_equal(orc.brief_describe_u8(big, [[50, 45]], 16, sp, cp)[0], inside)
    # border form: the word is shifted for every pair (unlike the F32 class), so out-of-image pairs leave 0 bits in place
    edge = orc.brief_describe_u8(img, [[3, 3]], 16, sp, cp)[0]
    for k in range(64):
        ax, ay = 3 + sp[cp[k, 0], 0], 3 + sp[cp[k, 0], 1]; bx, by = 3 + sp[cp[k, 1], 0], 3 + sp[cp[k, 1], 1]
        ok = 0 <= ax < 60 and 0 <= ay < 70 and 0 <= bx < 60 and 0 <= by < 70
        bit = (int(edge[k // 32]) >> (31 - k % 32)) & 1
        assert bit == (1 if ok and img[ay, ax] < img[by, bx] else 0)


def test_select_nbest_literals(orc):  # FT:alg/feature/detect/extract/TestSelectNBestFeatures.java:36-68 (testExtra), :73-93 (testTooLittle)
    a = np.zeros((20, 10), np.float32)
    a[10, 5] = -3; a[10, 4] = -3.5; a[11, 5] = 0; a[8, 8] = 10
    img = orc.Gray.from_array(a)
    corners = np.array([[5, 10], [4, 10], [5, 11], [8, 8]], np.int16)
    found = orc.select_nbest(img, corners, 3, positive=True)
    assert len(found) == 3 and tuple(found[0]) == (8, 8)
    assert {tuple(p) for p in found} == {(8, 8), (5, 11), (5, 10)}
    found = orc.select_nbest(img, corners, 3, positive=False)
    assert len(found) == 3 and tuple(found[0]) == (4, 10)
    # N larger than the list: an unpruned copy in the original order
    found = orc.select_nbest(img, corners, 20, positive=True)
    assert found.tolist() == corners.tolist()


def test_select_nbest_keeps_the_n_largest(orc):
    """Property the unpinned QuickSelect order cannot change: the kept set is the N largest intensities (no ties here), and
    FastHessianFeatureDetector with maxFeaturesPerScale = N returns a subset of the unlimited detection."""
    rand = orc.JavaRandom(99)
    img = rand.fillUniform(orc.Gray(64, 48), 0, 100)
    a = img.array()
    pts = np.array([[x, y] for y in range(0, 48, 3) for x in range(0, 64, 3)], np.int16)
    for n in (1, 7, 50, len(pts) - 1):
        kept = orc.select_nbest(img, pts, n)
        vals = sorted((a[y, x] for x, y in pts), reverse=True)
        assert sorted((a[y, x] for x, y in kept), reverse=True) == vals[:n]
    ii = orc.integral(orc.JavaRandom(5).fillUniform(orc.Gray(160, 120), 0, 100))
    full = {tuple(p) for p in orc.fh_detect(ii, orc.FhCfg()).tolist()}
    lim = orc.fh_detect(ii, orc.FhCfg(maxFeaturesPerScale=10)).tolist()
    assert 0 < len(lim) < len(full) and {tuple(p) for p in lim} <= full
