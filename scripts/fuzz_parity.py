#!/usr/bin/env python3
"""One-off wide parity sweep (GPU vs the CPU oracle), beyond what the test-suite runs every time: random frame shapes, detector
configurations, image statistics and batch positions for the Fast-Hessian detector (F32 and S32 integral images), the integral image
(both kernels), the Hessian intensity and SURF detect + describe.  Prints one line per mismatch and a summary; exit code 1 on any mismatch.

    python scripts/fuzz_parity.py [seed] [cases]"""
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


def frame(rng, w, h, kind):
    if kind == 0:
        return rng.uniform(0, 100, (h, w)).astype(np.float32)
    if kind == 1:
        return rng.integers(0, 256, (h, w)).astype(np.float32)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    a = np.full((h, w), 50.0)
    for _ in range(max(3, w * h // 3000)):
        cx, cy, sg = rng.uniform(0, w), rng.uniform(0, h), float(rng.choice([2, 3, 5, 8, 13]))
        a += float(rng.uniform(40, 100) * rng.choice([-1, 1])) * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * sg * sg))
    return (a + rng.uniform(0, 2, (h, w))).astype(np.float32)


def main(seed=None, cases=None):
    """seed / cases default to the command line (tests/test_gpu_fuzz_slice.py runs a bounded slice in-process)"""
    if seed is None:
        seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    if cases is None:
        cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    rng = np.random.default_rng(seed)
    orc.build()
    bad = 0
    points = 0
    t0 = time.time()
    for k in range(cases):
        w, h = int(rng.integers(40, 700)), int(rng.integers(40, 520))
        a = frame(rng, w, h, k % 3)
        img = orc.Gray.from_array(a)
        ii = orc.integral(img)
        cfg = {}
        r = k % 5
        if r == 1:
            cfg = dict(extractRadius=int(rng.integers(1, 4)), detectThreshold=float(rng.choice([0.5, 1.0, 10.0])))
        elif r == 2:
            cfg = dict(initialSampleSize=int(rng.integers(1, 4)), numberScalesPerOctave=int(rng.integers(3, 6)), numberOfOctaves=int(rng.integers(1, 5)),
                       initialSize=int(rng.choice([9, 15])), scaleStepSize=int(rng.choice([6, 8, 12])))
        elif r == 3:
            cfg = dict(maxFeaturesPerScale=int(rng.choice([3, 40, 400])))
        try:
            exp = orc.fh_detect(ii, orc.FhCfg(**cfg), threads=8)
            det = api.FastHessianFeatureDetector(api.ConfigFastHessian(**cfg))
            det.detect(G(ii))
            got = det.getFoundPoints()
            points += len(exp)
            if got.shape != exp.shape or not np.array_equal(got, exp):
                bad += 1
                print("MISMATCH fh_detect", w, h, cfg, got.shape, exp.shape, flush=True)
            if k % 4 == 0:
                skip, size = int(rng.choice([1, 2, 3, 4, 8])), int(rng.choice([9, 15, 21, 27, 39, 51]))
                if size < min(w, h):
                    out = api.GrayF32(w // skip, h // skip)
                    api.IntegralImageFeatureIntensity.hessian(G(ii), skip, size, out)
                    e = orc.hessian(ii, skip, size).array()
                    if not np.array_equal(out.array().view(np.uint32), e.view(np.uint32)):
                        bad += 1
                        print("MISMATCH hessian", w, h, skip, size, flush=True)
            if k % 6 == 0:
                stable = bool(k % 12)
                dd = (api.FactoryDetectDescribe.surfStable if stable else api.FactoryDetectDescribe.surfFast)(None, None, None, api.GrayF32)
                nb = int(rng.choice([1, 3, 130]))   # 130: the single-pass integral kernel
                dd.detectBatch([G(img)] * nb)
                ref = orc.Surf(stable)
                n = ref.detect(img)
                xys, ang, white, desc = ref.fetch()
                for pos in {0, nb - 1}:
                    dd.selectImage(pos)
                    g = dd._results()
                    ok = g[0].shape == xys.shape and np.array_equal(g[0], xys) and np.array_equal(g[2], white)
                    if ok and n:
                        derr = np.max(np.abs(g[3] - desc), axis=1)
                        ok = (derr <= 1e-5).mean() >= 0.995
                    if not ok:
                        bad += 1
                        print("MISMATCH surf", w, h, stable, nb, pos, flush=True)
                if not np.array_equal(dd.fetchIntegral(nb - 1, w, h).view(np.uint32), ii.array().view(np.uint32)):
                    bad += 1
                    print("MISMATCH integral", w, h, nb, flush=True)
            if k % 7 == 0:
                u8 = a.clip(0, 255).astype(np.uint8)
                iis = api.IntegralImageOps.transform(api.GrayU8.wrap(u8))
                fh = api.FastHessianFeatureDetector(api.ConfigFastHessian())
                fh.detect(iis)
                e = orc.fh_detect_s32(iis.array(), orc.FhCfg(), threads=8)
                if not np.array_equal(fh.getFoundPoints(), e):
                    bad += 1
                    print("MISMATCH fh_detect_s32", w, h, flush=True)
        except Exception as ex:   # a case the API rejects must be rejected by design, not crash: report it
            print("EXCEPTION", w, h, cfg, type(ex).__name__, str(ex)[:120], flush=True)
            bad += 1
        if k % 25 == 24:
            print("progress", k + 1, "cases", points, "key points", round(time.time() - t0, 1), "s, mismatches", bad, flush=True)
    print("done:", cases, "cases,", points, "key points,", bad, "mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
