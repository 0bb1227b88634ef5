"""Sharded greedy association over RCCL: one process per GPU, torch.distributed for the single exchange step.

Rank r owns source rows [begin_r, begin_r + n_r) of one global association problem and the whole destination set.
  phase 1 (local, GPU): forward matches of the local rows + per destination column the local (min1, min2, argmin1) record,
                        argmin1 as a GLOBAL source index                                   -> bhip_assoc_*_shard_phase1
  exchange            : ONE all-gather of the nd fixed-size records (24 B each; 16384 columns = 384 KiB per rank) -- latency bound,
                        xGMI bandwidth is irrelevant, so it is a single collective on the compute stream
  phase 2 (local, GPU): merge the R records of every matched column and keep (i -> m) iff i is the unique strict minimum of
                        column m (AssociateGreedy.java:105-114)                           -> bhip_assoc_shard_phase2
The forward pass needs no communication at all (each rank sees every destination).

The orchestration is independent of where phase 1/2 run: `GpuEngine` calls libboofhip.so on CUDA tensors (backend "nccl" == RCCL);
tests/test_sharded_gloo.py drives the same orchestration with world_size 2 on the "gloo" backend and an engine built from the
CPU oracle, which is test infrastructure only.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .api import Context, Double_MAX_VALUE, _check

RECORD_BYTES = 24  # struct ColTop {double min1, min2; int idx1, pad;}  == bhip_assoc_coltop_bytes()


def row_partition(n, world):
    """Contiguous, near-equal split of n source rows over `world` ranks: [(begin, count)] * world."""
    base, extra = divmod(n, world)
    out, b = [], 0
    for r in range(world):
        c = base + (1 if r < extra else 0)
        out.append((b, c))
        b += c
    return out


class GpuEngine:
    """phase 1 / phase 2 on the MI355X through the C ABI; every tensor is a CUDA tensor on the context's device."""

    def __init__(self, ctx=None, device=0):
        self.ctx = ctx or Context(device, stream=torch.cuda.current_stream(device).cuda_stream)
        self.L = _lib.load()
        assert self.L.bhip_assoc_coltop_bytes() == RECORD_BYTES
        self.device = torch.device("cuda", self.ctx.device)

    def phase1(self, kind, src, src_begin, dst, max_err):
        ns, nd = src.shape[0], dst.shape[0]
        pairs = torch.empty(max(ns, 1), dtype=torch.int32, device=self.device)
        fit = torch.empty(max(ns, 1), dtype=torch.float64, device=self.device)
        col = torch.empty(max(nd, 1) * RECORD_BYTES, dtype=torch.uint8, device=self.device)
        P = lambda t: C.c_void_p(t.data_ptr())
        if kind == "l2":
            st = self.L.bhip_assoc_l2_shard_phase1(self.ctx._h, P(src), ns, src_begin, P(dst), nd, src.shape[1], max_err, P(pairs), P(fit), P(col))
        else:
            st = self.L.bhip_assoc_hamming_shard_phase1(self.ctx._h, P(src), ns, src_begin, P(dst), nd, src.shape[1], max_err, P(pairs), P(fit), P(col))
        _check(self.ctx, st)
        return pairs[:ns], fit[:ns], col[:nd * RECORD_BYTES]

    def phase2(self, col_all, nranks, nd, pairs, fit, src_begin):
        P = lambda t: C.c_void_p(t.data_ptr())
        _check(self.ctx, self.L.bhip_assoc_shard_phase2(self.ctx._h, P(col_all), nranks, nd, pairs.shape[0], src_begin, P(pairs), P(fit)))
        return pairs, fit


def associate_sharded(engine, kind, src_local, src_begin, dst, max_err=Double_MAX_VALUE, backwards=True, group=None):
    """Greedy association of this rank's source rows against the full destination set.

    kind: "l2" (float64 descriptors, ScoreAssociateEuclideanSq_F64) or "hamming" (int32 words, ScoreAssociateHamming_B).
    Returns (pairs, fitQuality) for the local rows -- identical to the slice [src_begin : src_begin + n] of the single-GPU result.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    nd = dst.shape[0]
    pairs, fit, col = engine.phase1(kind, src_local, src_begin, dst, max_err)
    if not backwards or nd == 0:
        return pairs, fit
    if world > 1:
        col_all = torch.empty(world * col.numel(), dtype=torch.uint8, device=col.device)
        dist.all_gather_into_tensor(col_all, col.contiguous(), group=group)   # the one exchange step
    else:
        col_all = col
    return engine.phase2(col_all, world, nd, pairs, fit, src_begin)


def gather_matches(pairs, fit, counts, group=None):
    """Optional: assemble the full (pairs, fit) on every rank (ranks may own different numbers of rows)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return pairs, fit
    m = max(counts)
    pp = torch.full((m,), -1, dtype=pairs.dtype, device=pairs.device); pp[:pairs.numel()] = pairs
    ff = torch.zeros((m,), dtype=fit.dtype, device=fit.device); ff[:fit.numel()] = fit
    ap = torch.empty(world * m, dtype=pairs.dtype, device=pairs.device); af = torch.empty(world * m, dtype=fit.dtype, device=fit.device)
    dist.all_gather_into_tensor(ap, pp, group=group)
    dist.all_gather_into_tensor(af, ff, group=group)
    return (torch.cat([ap[r * m:r * m + counts[r]] for r in range(world)]), torch.cat([af[r * m:r * m + counts[r]] for r in range(world)]))
