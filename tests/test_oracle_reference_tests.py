"""More of the reference's own unit tests on the hot path, re-expressed against the CPU oracle (each one is an extra pin of the restatement
against the Java code: a shared misreading in oracle + product would have to survive these too).  Abbreviations as in SURVEY:
FT: = main/boofcv-feature/src/test/java/boofcv/, IT: = main/boofcv-ip/src/test/java/boofcv/, TT: = main/boofcv-types/src/test/java/boofcv/."""
import ctypes as C
import math

import numpy as np


# ---------------------------------------------------------------------------------------------------
# FT:alg/feature/describe/TestSurfDescribeOps.java:40-44 (60 x 70 image, centre), :116-146 isInside_aligned, :148-181 isInside_rotated
# ---------------------------------------------------------------------------------------------------
W, H = 60, 70
CX, CY = W // 2, H // 2


def _inside(orc, x, y, regionRadius, kernelSize, scale, c=0.0, s=0.0):
    return bool(orc.lib().orc_surf_is_inside(W, H, float(x), float(y), regionRadius, kernelSize, float(scale), float(c), float(s)))


def test_surf_is_inside_aligned(orc):
    regionRadius, kernelSize = 10, 2
    assert _inside(orc, CX, CY, regionRadius, kernelSize, 1)
    assert _inside(orc, CX, CY, regionRadius, kernelSize, 2)
    for swap in (False, True):   # lower boundary, in x then in y
        def check(x, y, scale, result):
            if swap:
                x, y = y, x
            assert _inside(orc, x, y, regionRadius, kernelSize, scale) == result, (x, y, scale, swap)
        check(regionRadius + 1 + 1, CY, 1, True)
        check(2 * regionRadius + 2 + 1, CY, 2, True)
        check(regionRadius + 1, CY, 1, False)
        check(2 * regionRadius + 2, CY, 2, False)
    # upper boundary
    assert _inside(orc, W - regionRadius - 1 - 1, CY, regionRadius, kernelSize, 1)
    assert _inside(orc, CX, H - regionRadius - 1 - 1, regionRadius, kernelSize, 1)
    assert not _inside(orc, W - regionRadius - 1, CY, regionRadius, kernelSize, 1)
    assert not _inside(orc, CX, H - regionRadius - 1, regionRadius, kernelSize, 1)


def test_surf_is_inside_rotated(orc):
    regionRadius, kernelSize = 10, 3
    d90, d45 = math.pi / 2.0, math.pi / 4.0
    fullRadius = regionRadius + (kernelSize // 2 + (kernelSize % 2) + 1)
    e = int(math.ceil(math.sqrt(2 * fullRadius * fullRadius))) - fullRadius
    assert _inside(orc, CX, CY, regionRadius, kernelSize, 1)
    assert _inside(orc, CX, CY, regionRadius, kernelSize, 2)
    assert _inside(orc, CX, CY, regionRadius, kernelSize, 1, math.cos(d90), math.sin(d90))
    assert _inside(orc, CX, CY, regionRadius, kernelSize, 2, math.cos(d90), math.sin(d90))
    assert _inside(orc, CX, CY, regionRadius, kernelSize, 1, math.cos(0.5), math.sin(0.5))
    # +2 = kernel radius, +1 = needing to sample x-1,y-1 below in the integral image
    assert not _inside(orc, regionRadius + 2, CY, regionRadius, kernelSize, 1)
    assert not _inside(orc, CX, regionRadius + 2, regionRadius, kernelSize, 1)
    assert not _inside(orc, W - regionRadius - 1 - 1, CY, regionRadius, kernelSize, 1)
    assert not _inside(orc, CX, H - regionRadius - 1 - 1, regionRadius, kernelSize, 1)
    assert _inside(orc, regionRadius + 2 + 1, CY, regionRadius, kernelSize, 1)
    assert _inside(orc, CX, regionRadius + 2 + 1, regionRadius, kernelSize, 1)
    assert _inside(orc, W - regionRadius - 2 - 1, CY, regionRadius, kernelSize, 1)
    assert _inside(orc, CX, H - regionRadius - 2 - 1, regionRadius, kernelSize, 1)
    # a rotation by 45 degrees needs `e` more pixels on every side (the rest of the Java test follows the same pattern)
    c, s = math.cos(d45), math.sin(d45)
    assert not _inside(orc, regionRadius + 2 + 1, CY, regionRadius, kernelSize, 1, c, s)
    assert _inside(orc, regionRadius + 2 + 1 + e, CY, regionRadius, kernelSize, 1, c, s)
    assert not _inside(orc, regionRadius + 2 + e, CY, regionRadius, kernelSize, 1, c, s)


# ---------------------------------------------------------------------------------------------------
# IT:factory/filter/kernel/TestFactoryKernelGaussian.java:53-182 ; IT:alg/filter/kernel/TestKernelMath.java:196-210,266-276
# ---------------------------------------------------------------------------------------------------
def _pdf(orc, x, sigma=1.0):
    return orc.lib().orc_compute_pdf(0.0, sigma, float(x))


def test_gaussian1d_f32_is_the_normalised_pdf(orc):
    k = orc.gaussian1d_f32(1.0, 2)   # FactoryKernelGaussian.gaussian(Kernel1D_F32.class, 1.0, 2) = odd width, normalised
    assert len(k) == 5
    norm = sum(_pdf(orc, i - 2) for i in range(5))
    for i in range(5):
        assert abs(k[i] - _pdf(orc, i - 2) / norm) < 1e-4
    assert abs(float(np.sum(k.astype(np.float64))) - 1.0) < 1e-4      # KernelMath.normalizeSumToOne
    assert abs(_pdf(orc, 0) - 1.0 / math.sqrt(2 * math.pi)) < 1e-15 and abs(_pdf(orc, 1.5, 2.0) - math.exp(-1.5 ** 2 / 8.0) / (2.0 * math.sqrt(2 * math.pi))) < 1e-15
    # radius from sigma and sigma from radius (FactoryKernelGaussian.radiusForSigma / sigmaForRadius, :297-322)
    assert len(orc.gaussian1d_f32(2.0, -1)) == 2 * int(math.ceil((5 * 2.0 - 1) / 2)) + 1
    assert np.array_equal(orc.gaussian1d_f32(-1, 3), orc.gaussian1d_f32((2 * 3 + 1) / 5.0, 3))


def test_gaussian2d_is_separable_and_sums_to_one(orc):
    k2 = orc.gaussian2d_f64(1.0, 2)
    assert k2.shape == (5, 5)
    k1 = np.array([_pdf(orc, i - 2) for i in range(5)])
    exp = np.outer(k1, k1)
    assert np.abs(k2 - exp / exp.sum()).max() < 1e-12 and abs(k2.sum() - 1) < 1e-12   # KernelMath.convolve2D(1D, 1D) + normalizeSumToOne


def _check_symmetry(a):
    w = a.shape[0]
    even = w % 2 == 0
    r = w // 2 - (1 if even else 0)
    for i in range(r + 1):
        for j in range(r + 1):
            assert abs(a[j, i] - a[j, w - 1 - i]) < 1e-8 and abs(a[j, i] - a[w - 1 - j, i]) < 1e-8
    if not even:
        assert abs(a[r, r] - a[r, r + 1]) > 1e-8 and abs(a[r, r] - a[r + 1, r]) > 1e-8


def test_gaussian_width_even_and_odd(orc):
    for width in (4, 5):
        for sigma in (2.0, -1.0):
            a = orc.gaussian_width(sigma, width)
            assert a.shape == (width, width)
            _check_symmetry(a)
            assert abs(a.sum() - 1) < 1e-8


# ---------------------------------------------------------------------------------------------------
# FT:alg/descriptor/TestUtilFeature.java:52-72 normalizeL2
# ---------------------------------------------------------------------------------------------------
def test_normalize_l2(orc):
    v = np.zeros(64)
    v[5], v[10] = 2, 4
    orc.lib().orc_normalize_l2(v.ctypes.data_as(C.POINTER(C.c_double)), 64)
    assert abs(v[5] - 0.44721) < 1e-3 and abs(v[10] - 0.89443) < 1e-3
    z = np.zeros(64)
    orc.lib().orc_normalize_l2(z.ctypes.data_as(C.POINTER(C.c_double)), 64)
    assert np.all(z == 0)    # the all-zero descriptor stays zero (no division by zero)


# ---------------------------------------------------------------------------------------------------
# IT:alg/transform/ii/TestDerivativeIntegralImage.java:147-279: integral kernels XX / YY / XY == dense convolution with the box kernels
# ---------------------------------------------------------------------------------------------------
def _deriv_xx(size):
    bw = size // 3
    bh = size - bw - 1
    by = (size - bh) // 2
    k = np.zeros((size, size))
    k[by:size - by, 0:bw] = 1
    k[by:size - by, 2 * bw:3 * bw] = 1
    k[by:size - by, bw:2 * bw] = -2
    return k


def _deriv_xy(size):
    b = size // 3
    border = (size - 2 * b - 1) // 2
    w = b * 3
    k = np.zeros((w, w))
    k[border:border + b, border:border + b] = 1
    k[border:border + b, border + b + 1:border + 2 * b + 1] = -1
    k[border + b + 1:size - border, border:border + b] = -1
    k[border + b + 1:size - border, border + b + 1:border + 2 * b + 1] = 1
    return k


def _correlate_zero_border(img, k):
    """ConvolveImage.convolve(kernel, orig, out, ImageBorderValue(0)): out(x,y) = sum k(i,j) * in(x + i - r, y + j - r)"""
    kh, kw = k.shape
    ry, rx = kh // 2, kw // 2
    p = np.pad(img.astype(np.float64), ((ry, ry), (rx, rx)))
    out = np.zeros(img.shape)
    for j in range(kh):
        for i in range(kw):
            if k[j, i] != 0:
                out += k[j, i] * p[j:j + img.shape[0], i:i + img.shape[1]]
    return out


def test_integral_derivative_kernels_equal_dense_convolution(orc):
    rand = orc.JavaRandom(234)
    orig = rand.fillUniform(orc.Gray(30, 40), 0, 20)
    ii = orc.integral(orig)
    L = orc.lib()
    for i in (1, 3, 5):
        size = i * 3
        for kind, kern in ((0, _deriv_xx(size)), (1, _deriv_xx(size).T), (2, _deriv_xy(size))):
            exp = _correlate_zero_border(orig.array(), kern)
            found = np.array([[L.orc_convolve_sparse(ii.c(), kind, size, x, y) for x in range(30)] for y in range(40)])
            assert np.abs(found - exp).max() < 1e-2, (size, kind)


# ---------------------------------------------------------------------------------------------------
# IT:alg/transform/ii/impl/TestSparseIntegralGradient_NoBorder_F32.java:33-55 (GeneralSparseGradientIntegralTests: sparse == dense convolution with
# DerivativeIntegralImage.kernelDerivX/Y(radius) inside the sample box bounds) ; TT:struct/deriv/TestSparseGradientSafe.java:38-52
# ---------------------------------------------------------------------------------------------------
def test_sparse_gradient_equals_box_derivative_and_is_safe_outside(orc):
    size, r = 5, 2
    rand = orc.JavaRandom(234)
    orig = rand.fillUniform(orc.Gray(20, 30), 0, 100)
    ii = orc.integral(orig)
    L = orc.lib()
    box = np.zeros(4, np.int32)
    L.orc_sparse_gradient_bounds(20, 30, float(size), 10, 10, box.ctypes.data_as(C.POINTER(C.c_int)))
    assert list(box) == [-r - 1, -r - 1, r, r]            # the sample box the test's constructor states
    kx = (np.array([-r - 1, -r - 1, -1, r, 0, -r - 1, r, r], np.int32), np.array([-1, 1], np.int32))   # DerivativeIntegralImage.kernelDerivX(r)
    ky = (np.array([-r - 1, -r - 1, r, -1, -r - 1, 0, r, r], np.int32), np.array([-1, 1], np.int32))   # kernelDerivY(r)
    I = C.POINTER(C.c_int)
    inside = 0
    for y in range(30):
        for x in range(20):
            gx, gy = C.c_float(), C.c_float()
            ok = L.orc_sparse_gradient(ii.c(), float(size), x, y, C.byref(gx), C.byref(gy))
            in_bounds = x - r - 1 >= 0 and y - r - 1 >= 0 and x + r < 20 and y + r < 30
            assert bool(ok) == in_bounds == bool(L.orc_sparse_gradient_bounds(20, 30, float(size), x, y, box.ctypes.data_as(I)))
            if not in_bounds:
                assert gx.value == 0 and gy.value == 0      # SparseGradientSafe: zero gradient instead of an exception
                continue
            inside += 1
            ex = L.orc_convolve_sparse_blocks(ii.c(), 2, kx[0].ctypes.data_as(I), kx[1].ctypes.data_as(I), x, y)
            ey = L.orc_convolve_sparse_blocks(ii.c(), 2, ky[0].ctypes.data_as(I), ky[1].ctypes.data_as(I), x, y)
            assert abs(gx.value - ex) < 1e-2 * max(1.0, abs(ex)) and abs(gy.value - ey) < 1e-2 * max(1.0, abs(ey)), (x, y)
    assert inside == (20 - 2 * r - 1) * (30 - 2 * r - 1)


# ---------------------------------------------------------------------------------------------------
# FT:alg/feature/associate/TestFindUnassociated.java + FT:abst/feature/associate/TestWrapAssociateGreedy.java (StandardAssociateDescriptionChecks with
# ScoreAssociateEuclidean_F64, backwards off and on) on the ORACLE's greedy: the lists WrapAssociateGreedy derives from pairs[]
# ---------------------------------------------------------------------------------------------------
def _wrap(pairs, fit, nd):
    matches = [(i, int(p), float(fit[i])) for i, p in enumerate(pairs) if p >= 0]
    unsrc = [i for i, p in enumerate(pairs) if p < 0]
    matched = {m[1] for m in matches}
    undst = [j for j in range(nd) if j not in matched]
    return matches, unsrc, undst


def test_wrap_associate_greedy_standard_checks_on_the_oracle(orc):
    for backwards in (False, True):
        # basic(): 20 features, unique pairs i <-> i+0.001*..., threshold 0.01 (StandardAssociateDescriptionChecks.java:76-105)
        src = np.arange(20, dtype=np.float64).reshape(-1, 1) * 10
        dst = src + 1.0
        p, f = orc.associate_l2(src, dst, 0.01, backwards, sqrt_score=True)
        assert _wrap(p, f, 20)[0] == []                       # nothing within 0.01
        p, f = orc.associate_l2(src, dst, 1.0 + 1e-9, backwards, sqrt_score=True)
        m, us, ud = _wrap(p, f, 20)
        assert [(a, b) for a, b, _ in m] == [(i, i) for i in range(20)] and us == [] and ud == []
        assert all(abs(q - 1.0) < 1e-12 for _, _, q in m)     # Euclidean (sqrt) score
        # checkUnassociated*(): 2 matches, 1 unassociated source, 2 unassociated destinations (:170-240)
        src = np.array([[1.0], [2.0], [100.0]])
        dst = np.array([[1.0], [2.0], [50.0], [300.0]])
        p, f = orc.associate_l2(src, dst, 5.0, backwards, sqrt_score=True)
        m, us, ud = _wrap(p, f, 4)
        assert [(a, b) for a, b, _ in m] == [(0, 0), (1, 1)] and us == [2] and ud == [2, 3]
        # the threshold is inclusive (:133-150 uses 1.1 - 1 as the cut)
        src = np.array([[1.0]]); dst = np.array([[1.1]])
        d = math.sqrt((1.0 - 1.1) ** 2)
        assert _wrap(*orc.associate_l2(src, dst, d, backwards, sqrt_score=True), 1)[0] != []
        assert _wrap(*orc.associate_l2(src, dst, np.nextafter(d, 0), backwards, sqrt_score=True), 1)[0] == []
    # uniqueDestination only with backwards validation: two sources closest to the same destination
    src = np.array([[1.0], [1.2]]); dst = np.array([[1.1], [9.0]])
    p, _ = orc.associate_l2(src, dst, 10.0, False, sqrt_score=True)
    assert list(p) == [0, 0]
    p, _ = orc.associate_l2(src, dst, 10.0, True, sqrt_score=True)
    assert list(p).count(0) <= 1
