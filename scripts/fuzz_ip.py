#!/usr/bin/env python3
"""One-off wide parity sweep of the boofcv-ip front end (GPU vs the CPU oracle), bit-exact: separable convolution (plain and
border-normalised, unrolled and standard widths), Gaussian blur, mean / median blur, Sobel / three-tap gradients (with and without the
zero border), the down-sampling convolution and the discrete pyramid (incl. the step >= 3 quirks), Shi-Tomasi / Harris corner
intensity -- random shapes, radii, steps and value ranges.

    python scripts/fuzz_ip.py [seed] [cases]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from boofcv_amd import api                 # noqa: E402
from oracle import pyoracle as orc         # noqa: E402  (checker)


def G(img):
    return api.GrayF32.wrap(img.array())


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def main(seed=None, cases=None):
    """seed / cases default to the command line (tests/test_gpu_fuzz_slice.py runs a bounded slice in-process)"""
    if seed is None:
        seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    if cases is None:
        cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    rng = np.random.default_rng(seed)
    orc.build()
    bad = 0
    checks = 0
    t0 = time.time()

    def check(name, got, exp, *ctx):
        nonlocal bad, checks
        checks += 1
        if got.shape != exp.shape or not np.array_equal(bits(got), bits(exp)):
            bad += 1
            print("MISMATCH", name, *ctx, flush=True)

    for k in range(cases):
        w, h = int(rng.integers(12, 400)), int(rng.integers(12, 300))
        lo, hi = [(0, 50), (-5, 5), (0, 255), (1e3, 1e4)][k % 4]
        img = orc.Gray.from_array(rng.uniform(lo, hi, (h, w)).astype(np.float32))
        try:
            r = int(rng.integers(1, 9))
            ker = orc.gaussian1d_f32(-1, r)
            for kind, cls, fn in [("h", api.ConvolveImageNoBorder, "horizontal"), ("v", api.ConvolveImageNoBorder, "vertical"),
                                  ("norm_h", api.ConvolveImageNormalized, "horizontal"), ("norm_v", api.ConvolveImageNormalized, "vertical")]:
                if not kind.startswith("norm") and len(ker) > min(w, h):
                    continue
                exp = orc.conv(kind, ker, r, img).array()
                out = api.GrayF32(w, h)
                getattr(cls, fn)(api.Kernel1D_F32(ker), G(img), out)
                check("conv " + kind, out.array(), exp, w, h, r)
            if 2 * r + 1 <= 2 * min(w, h):
                check("gaussian", api.BlurImageOps.gaussian(G(img), None, -1, r).array(), orc.gaussian_blur(img, -1, r).array(), w, h, r)
            rm = int(rng.integers(1, 5))
            if 2 * rm + 1 <= min(w, h):
                check("mean", api.BlurImageOps.mean(G(img), None, rm).array(), orc.blur_mean(img, rm).array(), w, h, rm)
                if rm <= 3:
                    check("median", api.BlurImageOps.median(G(img), None, rm).array(), orc.blur_median(img, rm).array(), w, h, rm)
            for gk, cls in [("sobel", api.GradientSobel), ("three", api.GradientThree)]:
                dx, dy = api.GrayF32(w, h), api.GrayF32(w, h)
                cls.process(G(img), dx, dy, 0)
                ex, ey = orc.gradient(gk, img, border_zero=True)
                check(gk + " dx", dx.array(), ex.array(), w, h); check(gk + " dy", dy.array(), ey.array(), w, h)
                if gk == "sobel":
                    rc = int(rng.integers(1, 4))
                    if 2 * rc + 1 <= min(w, h):
                        for alg, name, kappa in [(api.FactoryIntensityPointAlg.shiTomasi(rc, False, api.GrayF32), "shitomasi", 0.0),
                                                 (api.FactoryIntensityPointAlg.harris(rc, 0.04, False, api.GrayF32), "harris", 0.04)]:
                            inten = api.GrayF32(1, 1)
                            alg.process(dx, dy, inten)
                            check("corner " + name, inten.array(), orc.corner_intensity(ex, ey, rc, name, kappa), w, h, rc)
            # down-sampling convolution, one direction, random step
            skip = int(rng.integers(1, 6))
            kd = orc.gaussian1d_f32(-1, int(rng.integers(1, 6)))
            for kind in ("h", "v"):
                ow, oh = (w // skip, h) if kind == "h" else (w, h // skip)
                if ow == 0 or oh == 0:
                    continue
                exp = orc.Gray(ow, oh); exp.buf[:] = -3.0
                fn = api.ConvolveImageDownNormalized.horizontal if kind == "h" else api.ConvolveImageDownNormalized.vertical
                out = api.GrayF32(ow, oh); out.data[:] = -3.0
                try:
                    orc.conv_down(kind, kd, img, skip, out=exp)
                except ValueError:
                    try:
                        fn(api.Kernel1D_F32(kd), G(img), out, skip)
                        bad += 1
                        print("MISSING REJECTION conv_down", kind, w, h, len(kd), skip, flush=True)
                    except api.IllegalArgumentException:
                        checks += 1
                    continue
                fn(api.Kernel1D_F32(kd), G(img), out, skip)
                check("conv_down " + kind, out.array(), exp.array(), w, h, len(kd), skip)
            # pyramid
            if k % 3 == 0:
                s0 = int(rng.choice([1, 1, 2]))
                scales = [s0]
                while len(scales) < int(rng.integers(2, 5)):
                    scales.append(scales[-1] * int(rng.choice([1, 2, 2, 3])))
                rp = int(rng.integers(1, 4))
                kp = orc.gaussian1d_f32(-1, rp)
                try:
                    exp_layers, _ = orc.pyramid(kp, -1, scales, img)
                except ValueError:
                    exp_layers = None
                try:
                    pyr = api.FactoryPyramid.discreteGaussian(scales, -1, rp)
                    pyr.process(G(img))
                    got_layers = [pyr.getLayer(i).array() for i in range(len(scales))]
                except (api.IllegalArgumentException, RuntimeError):
                    got_layers = None
                if (exp_layers is None) != (got_layers is None):
                    bad += 1
                    print("MISMATCH pyramid acceptance", w, h, scales, rp, exp_layers is None, got_layers is None, flush=True)
                elif exp_layers is not None:
                    for i, e in enumerate(exp_layers):
                        check("pyramid layer %d" % i, got_layers[i], e, w, h, scales, rp)
        except Exception as ex:
            bad += 1
            print("EXCEPTION", w, h, type(ex).__name__, str(ex)[:160], flush=True)
        if k % 25 == 24:
            print("progress", k + 1, "cases", checks, "checks", round(time.time() - t0, 1), "s, mismatches", bad, flush=True)
    print("done:", cases, "cases,", checks, "checks,", bad, "mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
