"""world_size-2 `gloo` test of the sharded association orchestration (boofcv_amd/sharded.py) on CPU.

The phase-1 / phase-2 engine here is built from the CPU oracle (test infrastructure) so the collective plumbing -- row partition,
record layout, the single all-gather, global source indices, the merge rule -- runs without a GPU.  The same orchestration with the
GPU engine is covered by tests/test_gpu_parity.py::test_sharded_association_single_process_ranks.
"""
import os
import struct
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class OracleEngine:
    """phase1/phase2 with the semantics of bhip_assoc_*_shard_phase1/2, computed on the CPU from exact scores."""

    def __init__(self, orc):
        self.orc = orc

    def _scores(self, kind, src, dst):
        s, d = src.numpy(), dst.numpy()
        if kind == "l2":
            out = np.zeros((len(s), len(d)))
            for i in range(len(s)):
                diff = s[i][None, :] - d
                acc = np.zeros(len(d))
                for k in range(s.shape[1]):  # sequential sum, as DescriptorDistance.euclideanSq
                    acc = acc + diff[:, k] * diff[:, k]
                out[i] = acc
            return out
        x = np.bitwise_xor(s[:, None, :].astype(np.uint32), d[None, :, :].astype(np.uint32))
        return np.unpackbits(x.view(np.uint8), axis=-1).sum(axis=-1).astype(np.float64)

    def phase1(self, kind, src, src_begin, dst, max_err):
        W = self._scores(kind, src, dst)
        ns, nd = W.shape
        pairs = np.full(ns, -1, np.int32); fit = np.full(ns, max_err)
        for i in range(ns):
            best, idx = max_err, -1
            for j in range(nd):
                if W[i, j] <= best:
                    best, idx = W[i, j], j
            pairs[i], fit[i] = idx, best
        rec = bytearray()
        for j in range(nd):
            col = W[:, j]
            m1, m2, i1 = np.inf, np.inf, -1
            for i in range(ns):
                v = col[i]
                if v < m1:
                    m2, m1, i1 = m1, v, src_begin + i
                elif v < m2:
                    m2 = v
            rec += struct.pack("<ddii", m1, m2, i1, 0)
        return torch.from_numpy(pairs), torch.from_numpy(fit), torch.frombuffer(bytearray(rec), dtype=torch.uint8).clone()

    def phase2(self, col_all, nranks, nd, pairs, fit, src_begin):
        raw = col_all.numpy().tobytes()
        recs = [struct.unpack_from("<ddii", raw, 24 * k) for k in range(nranks * nd)]
        pairs = pairs.clone(); fit = fit.clone()
        for i in range(len(pairs)):
            m = int(pairs[i])
            if m < 0:
                continue
            m1, m2, i1 = np.inf, np.inf, -1
            for r in range(nranks):
                a1, a2, ai, _ = recs[r * nd + m]
                if a1 < m1:
                    m2 = min(m1, a2); m1 = a1; i1 = ai
                elif a1 < m2:
                    m2 = a1
            if not (i1 == src_begin + i and m2 > m1):
                pairs[i] = -1; fit[i] = 1.7976931348623157e308
        return pairs, fit


def _worker(rank, world, port, kind, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pyoracle as orc
        from boofcv_amd import sharded
        rng = np.random.default_rng(42)  # same data on every rank
        ns, nd = 37, 29
        if kind == "l2":
            src = rng.normal(size=(ns, 8)); dst = rng.normal(size=(nd, 8))
            dst[:10] = src[5:15] + rng.normal(scale=0.01, size=(10, 8)); src[30] = src[7]; dst[20] = dst[3]  # duplicates -> ties
            full_p, full_f = orc.associate_l2(src, dst, 1e300, True)
            ts, td = torch.from_numpy(src), torch.from_numpy(dst)
        else:
            src = rng.integers(-2**31, 2**31, size=(ns, 4), dtype=np.int64).astype(np.int32)
            dst = rng.integers(-2**31, 2**31, size=(nd, 4), dtype=np.int64).astype(np.int32)
            dst[:12] = src[3:15]; dst[1] ^= 5; src[33] = src[4]
            full_p, full_f = orc.associate_hamming(src, dst, 1e300, True)
            ts, td = torch.from_numpy(src), torch.from_numpy(dst)
        part = sharded.row_partition(ns, world)
        b, c = part[rank]
        p, f = sharded.associate_sharded(OracleEngine(orc), kind, ts[b:b + c], b, td, 1e300, True)
        ok = np.array_equal(p.numpy(), full_p[b:b + c]) and np.array_equal(f.numpy(), full_f[b:b + c])
        allp, allf = sharded.gather_matches(p, f, [cc for _, cc in part])
        ok = ok and np.array_equal(allp.numpy(), full_p) and np.array_equal(allf.numpy(), full_f)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["l2", "hamming"])
def test_sharded_association_two_ranks_gloo(kind, orc):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 1000) + (0 if kind == "l2" else 1000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    results = dict(q.get(timeout=5) for _ in range(world))
    assert results == {0: True, 1: True}


def test_row_partition():
    from boofcv_amd import sharded
    assert sharded.row_partition(10, 4) == [(0, 3), (3, 3), (6, 2), (8, 2)]
    assert sharded.row_partition(3, 8)[3] == (3, 0)
    assert sum(c for _, c in sharded.row_partition(16384, 8)) == 16384
