#!/usr/bin/env python3
"""Writes tests/golden/config1_small.npz and config1_640x480.npz: frozen runs of BASELINE config 1 (two S-noise frames through surfStable,
greedy mutual-best Euclidean-squared association) -- scaled down to 320x240, and at the configuration's real 640x480 (there every
8th descriptor is stored, plus the row sum of every descriptor, to keep the file small).

IMPORTANT: the vectors come from oracle/ -- the C++ restatement of the reference's Java code -- NOT from BoofCV itself.  This image has no
JVM and the reference ships no stored outputs, so these fixtures cannot pin parity with the Java build; they pin the
restatement (a change in the oracle's arithmetic shows up here) and give the GPU tests a target that does not depend on the oracle being
rebuilt.  Parity with Java rests on the reference's own known-answer literals (tests/test_oracle_known_answers.py) plus construction.

Inputs are regenerated from the seeds (ImageMiscOps.fillUniform with java.util.Random, restated in oracle/), so only outputs are stored:
key points (x, y, scale: float64, exact), orientation, Laplacian sign, descriptors as float32 (the parity bar is 1e-5), association
pairs and fit scores, BRIEF-512 words at the first 64 key points, and -- for an exact association check -- every 8th descriptor of both
frames as float64 together with the association of those two subsets.

    python tests/golden/make_golden.py          # rewrites the fixture (review the diff!)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

W, H, SEEDS = 320, 240, (234, 235)
FULL_W, FULL_H = 640, 480     # BASELINE config 1 at its real size (config1_640x480.npz)
FULL_DESC_EVERY = 8           # the full-size fixture stores every 8th descriptor (+ a row sum of every descriptor) to stay small


def generate(width=W, height=H, desc_every=1):
    from oracle import pyoracle as orc
    orc.build()
    out = {}
    descs = []
    for k, seed in enumerate(SEEDS):
        img = orc.noise_image(width, height, seed)
        s = orc.Surf(True)
        n = s.detect(img)
        xys, ang, white, desc = s.fetch()
        out["xys%d" % k] = xys
        out["angle%d" % k] = ang
        out["white%d" % k] = white.astype(np.uint8)
        out["desc%d" % k] = desc[::desc_every].astype(np.float32)
        if desc_every > 1:
            out["descsum%d" % k] = desc.sum(axis=1)      # float64 row sums of ALL descriptors: a cheap check of the ones not stored
        descs.append(desc)
        if k == 0:
            sp, cp = orc.brief_definition()
            out["brief0"] = orc.brief_describe(img, xys[:64, :2], 16, sp, cp)
    pairs, fit = orc.associate_l2(descs[0], descs[1], backwards=True)
    out["pairs"] = pairs
    out["fit"] = fit
    # An association whose inputs are stored exactly: every 8th descriptor of both frames as float64, and the oracle's greedy mutual-best
    # association of those two subsets.  A GPU association of these stored inputs must reproduce pairs64 / fit64 bit for bit (the full-set
    # comparison above goes through the GPU's own descriptors, which differ from the oracle's in the last bits).
    out["desc64_0"] = np.ascontiguousarray(descs[0][::8])
    out["desc64_1"] = np.ascontiguousarray(descs[1][::8])
    out["pairs64"], out["fit64"] = orc.associate_l2(out["desc64_0"], out["desc64_1"], backwards=True)
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    for name, args in (("config1_small.npz", (W, H, 1)), ("config1_640x480.npz", (FULL_W, FULL_H, FULL_DESC_EVERY))):
        data = generate(*args)
        path = os.path.join(here, name)
        np.savez_compressed(path, **data)
        print(path, {k: v.shape for k, v in data.items()}, os.path.getsize(path), "bytes")
