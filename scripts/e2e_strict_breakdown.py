#!/usr/bin/env python3
"""Where the STRICT host-boundary time goes (bench.py end_to_end.value: the reference's own per-call interface): bhip_surf_detect_f32 in
sub-batches of 32 pinned frames, bhip_surf_fetch per frame, bhip_assoc_l2_f64 per consecutive pair -- one host thread, so the pieces add up.
Diagnostic only."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from boofcv_amd import api  # noqa: E402

B, H, W, SUB = 128, 1080, 1920, 32
dev = torch.device("cuda", 0)
frames = bench.synth_frames(B, H, W, 1000, dev)
host = torch.empty((B, H, W), dtype=torch.float32, pin_memory=True)
host.copy_(frames); torch.cuda.synchronize()
fr = host.numpy()
ctx = api.Context(0)
dd = api.FactoryDetectDescribe.surfStable(None, None, None, api.GrayF32, ctx=ctx)
assoc = api.FactoryAssociation.greedy(api.ScoreAssociateEuclideanSq_F64(), api.Double_MAX_VALUE, True, ctx=ctx)
for rep in range(3):
    t_det = t_fetch = t_assoc = 0.0
    descs = [None] * B
    for a in range(0, B, SUB):
        imgs = [api.GrayF32(W, H, fr[i].reshape(-1)) for i in range(a, a + SUB)]
        t0 = time.perf_counter(); dd.detectBatch(imgs); t1 = time.perf_counter()
        for j in range(SUB):
            descs[a + j] = dd._results(j)[3]
        t2 = time.perf_counter()
        t_det += t1 - t0; t_fetch += t2 - t1
    t0 = time.perf_counter()
    for i in range(B):
        assoc.setSource(descs[i]); assoc.setDestination(descs[(i + 1) % B]); assoc.associate()
    t_assoc = time.perf_counter() - t0
    tot = t_det + t_fetch + t_assoc
    print("rep %d (%d frames, one thread): detect %.2f ms/frame  fetch %.2f ms/frame  associate %.2f ms/pair  total %.2f ms/frame = %.0f frames/s" %
          (rep, B, 1e3 * t_det / B, 1e3 * t_fetch / B, 1e3 * t_assoc / B, 1e3 * tot / B, B / tot), flush=True)
# single-frame calls (the reference's detect(T) as written): latency of one pass through the launch sequence
img = api.GrayF32(W, H, fr[0].reshape(-1))
for _ in range(3):
    dd.detect(img)
t0 = time.perf_counter()
for _ in range(20):
    dd.detect(img); dd._results(0)
print("single-frame detect + fetch: %.2f ms per call" % (1e3 * (time.perf_counter() - t0) / 20))
