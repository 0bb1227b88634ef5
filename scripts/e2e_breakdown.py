#!/usr/bin/env python3
"""Where the batch-level host-boundary time goes (bench.py end_to_end.batched_calls): detectBatch (H2D + kernels), fetchAll (D2H),
associateImages, on 256 pinned 1080p frames.  Diagnostic only."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from boofcv_amd import api  # noqa: E402

B, H, W = 256, 1080, 1920
dev = torch.device("cuda", 0)
frames = bench.synth_frames(B, H, W, 1000, dev)
host = torch.empty((B, H, W), dtype=torch.float32, pin_memory=True)
host.copy_(frames); torch.cuda.synchronize()
fr = host.numpy()
ctx = api.Context(0)
dd = api.FactoryDetectDescribe.surfStable(None, None, None, api.GrayF32, ctx=ctx)
for rep in range(3):
    t0 = time.perf_counter()
    imgs = [api.GrayF32(W, H, fr[i].reshape(-1)) for i in range(B)]
    t1 = time.perf_counter()
    dd.detectBatch(imgs)
    t2 = time.perf_counter()
    out = dd.fetchAll()
    t3 = time.perf_counter()
    src = np.arange(B, dtype=np.int32)
    pairs, fit = dd.associateImages(src, (src + 1) % B)
    t4 = time.perf_counter()
    print("rep %d: wrap %.1f ms  detectBatch %.1f ms  fetchAll %.1f ms  associateImages %.1f ms  total %.1f ms  (%d key points)" %
          (rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), 1e3 * (t4 - t0), int(out[4][-1])), flush=True)
# plain pinned H2D of the same bytes, for scale
d = torch.empty((B, H, W), dtype=torch.float32, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter(); d.copy_(host, non_blocking=True); torch.cuda.synchronize()
print("torch pinned H2D of the batch: %.1f ms" % (1e3 * (time.perf_counter() - t0)))
