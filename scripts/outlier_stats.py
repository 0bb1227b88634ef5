#!/usr/bin/env python3
"""One-off diagnostic: how far are GPU orientations / SURF descriptors from the oracle's, and is every descriptor outside 1e-5 explained by
a last-bits orientation difference (the oracle, given the GPU's angle, reproduces the GPU's descriptor)?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from boofcv_amd import api
from oracle import pyoracle as orc
from test_gpu_fullsize import blobs

T = min(os.cpu_count() or 1, 16)
cases = [("noise 640x480 s234", orc.noise_image(640, 480, 234)), ("noise 400x300 s6", orc.noise_image(400, 300, 6)), ("noise 1920x1080 s7", orc.noise_image(1920, 1080, 7)),
         ("blobs 1920x1080 s1000", blobs(orc, 1920, 1080, 1000)), ("blobs 3840x2160 s5000", blobs(orc, 3840, 2160, 5000)), ("blobs 1920x1080 s1001", blobs(orc, 1920, 1080, 1001))]
for stable in (True, False):
    dd = (api.FactoryDetectDescribe.surfStable if stable else api.FactoryDetectDescribe.surfFast)(None, None, None, api.GrayF32)
    ref = orc.Surf(stable)
    for name, img in cases:
        dd.detect(api.GrayF32(img.width, img.height, img.buf))
        n = ref.detect(img, threads=T)
        xys, ang, white, desc = ref.fetch()
        got = dd._results()
        assert np.array_equal(got[0], xys)
        dang = np.abs(np.angle(np.exp(1j * (got[1] - ang))))
        derr = np.max(np.abs(got[3] - desc), axis=1)
        out = np.nonzero(derr > 1e-5)[0]
        ii = orc.Gray.from_array(ref.integral())
        expl = []
        for k in out:
            d2, _ = orc.describe(ii, xys[k, 0], xys[k, 1], got[1][k], xys[k, 2], stable)
            expl.append((int(k), float(dang[k]), float(derr[k]), float(np.max(np.abs(d2 - got[3][k])))))
        print("stable=%s %-24s n=%6d  max|dang|=%.3g  #dang>1e-12=%d  #dang>1e-9=%d  max derr (inliers)=%.3g  outliers=%s" % (
            stable, name, n, dang.max(), int((dang > 1e-12).sum()), int((dang > 1e-9).sum()), derr[derr <= 1e-5].max(), expl), flush=True)
