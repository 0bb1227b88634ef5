#!/usr/bin/env python3
"""Side measurement of the boofcv-ip front end on 1080p frames (host buffers -> the numbers are upload-inclusive) and, via
bench_parts.py 5, the device-resident pyramid.  Prints kernel ms from the ctx profiler for one Gaussian blur and one Sobel pass."""
import os, sys, json, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from boofcv_amd import api


def main():
    ctx = api.Context.default()
    rng = np.random.default_rng(1)
    img = api.GrayF32.wrap(rng.uniform(0, 255, (2160, 3840)).astype(np.float32))
    out = api.GrayF32(3840, 2160)
    for radius in (2, 5, 20):
        ctx.profile(True); ctx.profileReset()
        for _ in range(3):
            api.BlurImageOps.gaussian(img, out, -1, radius, None)
        ctx.synchronize()
        prof = ctx.profileReport(); ctx.profile(False)
        print(json.dumps({"op": "gaussian r=%d 3840x2160" % radius, "kernels_ms": {k: round(v["ms"] / 3, 4) for k, v in prof.items()}}))


if __name__ == "__main__":
    main()


def corner_and_mean():
    """Shi-Tomasi corner intensity, mean and median blur on one 1080p frame: kernel ms from the ctx profiler."""
    ctx = api.Context.default()
    rng = np.random.default_rng(2)
    w, h = 1920, 1080
    img = api.GrayF32.wrap(rng.uniform(0, 255, (h, w)).astype(np.float32))
    dx, dy = api.GrayF32(w, h), api.GrayF32(w, h)
    api.GradientSobel.process(img, dx, dy, 0)
    alg = api.FactoryIntensityPointAlg.shiTomasi(2, False, api.GrayF32)
    inten = api.GrayF32(1, 1)
    out = api.GrayF32(w, h)
    for name, fn in (("shiTomasi r=2", lambda: alg.process(dx, dy, inten)), ("mean r=3", lambda: api.BlurImageOps.mean(img, out, 3, None)),
                     ("median r=2", lambda: api.BlurImageOps.median(img, out, 2))):
        fn()
        ctx.profile(True); ctx.profileReset()
        for _ in range(3):
            fn()
        ctx.synchronize()
        prof = ctx.profileReport(); ctx.profile(False)
        print(json.dumps({"op": name + " 1920x1080", "kernels_ms": {k: round(v["ms"] / 3, 4) for k, v in prof.items()}}))


if __name__ == "__main__":
    corner_and_mean()
