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
