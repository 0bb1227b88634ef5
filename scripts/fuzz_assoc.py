#!/usr/bin/env python3
"""One-off wide parity sweep of the association paths (GPU vs the CPU oracle): greedy L2 (the fp32-MFMA candidate search with exact
fp64 re-scoring, and the exact VALU path for dof != 64) and Hamming (int8-MFMA for 512 bits, VALU otherwise), with the input classes the
exactness argument has to survive: near-duplicates at graded distances (1e-3 ... 1e-9, i.e. inside and far inside the fp32 candidate
band), exact duplicates (ties: last index wins forward, ties invalidate backward), scaled descriptors (large norms), thresholds that cut
through the score distribution, ragged sizes.  Pairs and fit scores must be identical.

    python scripts/fuzz_assoc.py [seed] [cases]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from boofcv_amd import api                 # noqa: E402
from oracle import pyoracle as orc         # noqa: E402  (checker)


def greedy(score, maxErr, backwards, src, dst):
    alg = api.FactoryAssociation.greedy(score, maxErr, backwards)
    alg.setSource(src); alg.setDestination(dst); alg.associate()
    return alg.getPairs(), alg.getFitQuality()


def main(seed=None, cases=None):
    """seed / cases default to the command line (tests/test_gpu_fuzz_slice.py runs a bounded slice in-process)"""
    if seed is None:
        seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    if cases is None:
        cases = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    rng = np.random.default_rng(seed)
    orc.build()
    bad = 0
    t0 = time.time()
    for k in range(cases):
        backwards = bool(rng.integers(0, 2))
        if k % 3 != 2:
            dof = 64 if k % 6 != 4 else int(rng.choice([1, 7, 32, 128]))
            ns, nd = int(rng.integers(1, 1500)), int(rng.integers(1, 1500))
            src = rng.normal(size=(ns, dof)); src /= np.linalg.norm(src, axis=1, keepdims=True)
            dst = rng.normal(size=(nd, dof)); dst /= np.linalg.norm(dst, axis=1, keepdims=True)
            m = min(ns, nd)
            # graded near-duplicates, exact duplicates, a scaled block
            for lo, hi, eps in [(0, m // 4, 10.0 ** -rng.integers(3, 10)), (m // 4, m // 2, 0.05)]:
                if hi > lo:
                    dst[lo:hi] = src[lo:hi] + rng.normal(scale=eps, size=(hi - lo, dof))
            if m >= 8:
                dst[rng.integers(0, nd, 4)] = dst[rng.integers(0, nd, 4)]       # duplicate destinations
                src[rng.integers(0, ns, 4)] = src[rng.integers(0, ns, 4)]       # duplicate sources
            scale = float(rng.choice([1.0, 1.0, 255.0, 1e-3]))
            src *= scale; dst *= scale
            maxErr = float(rng.choice([api.Double_MAX_VALUE, 0.5 * scale * scale, 1e-6 * scale * scale, 1.9 * scale * scale]))
            p, f = greedy(api.ScoreAssociateEuclideanSq_F64(), maxErr, backwards, src, dst)
            ep, ef = orc.associate_l2(src, dst, maxErr, backwards, threads=8)
            if not (np.array_equal(p, ep) and np.array_equal(f, ef)):
                bad += 1
                print("MISMATCH l2", ns, nd, dof, backwards, maxErr, scale, int((p != ep).sum()), flush=True)
        else:
            words = 16 if k % 2 else int(rng.choice([1, 3, 8, 32]))
            ns, nd = int(rng.integers(1, 2500)), int(rng.integers(1, 2500))
            src = rng.integers(-2 ** 31, 2 ** 31, (ns, words), dtype=np.int64).astype(np.int32)
            dst = rng.integers(-2 ** 31, 2 ** 31, (nd, words), dtype=np.int64).astype(np.int32)
            m = min(ns, nd)
            if m >= 4:
                dst[: m // 2] = src[: m // 2]
                flips = rng.integers(0, 32 * words, (m // 2, 6))
                for t in range(6):   # up to six bit flips per copied row (some rows get none: exact duplicates / ties)
                    rows = np.nonzero(rng.random(m // 2) < 0.6)[0]
                    bitpos = flips[rows, t]
                    dst[rows, bitpos // 32] ^= np.left_shift(np.uint32(1), (bitpos % 32).astype(np.uint32)).view(np.int32)
            maxErr = float(rng.choice([api.Double_MAX_VALUE, 3.0, 0.0, 16.0 * words]))
            p, f = greedy(api.ScoreAssociateHamming_B(), maxErr, backwards, src, dst)
            ep, ef = orc.associate_hamming(src, dst, maxErr, backwards, threads=8)
            if not (np.array_equal(p, ep) and np.array_equal(f, ef)):
                bad += 1
                print("MISMATCH hamming", ns, nd, words, backwards, maxErr, int((p != ep).sum()), flush=True)
        if k % 25 == 24:
            print("progress", k + 1, "cases", round(time.time() - t0, 1), "s, mismatches", bad, flush=True)
    print("done:", cases, "cases,", bad, "mismatches")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
