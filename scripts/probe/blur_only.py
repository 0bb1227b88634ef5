"""Gaussian blur r = 5 (and r = 2) on a resident 64 x 1080p batch, a few launches: the workload of scripts/pmc_blur.sh."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from boofcv_amd import api, device as dv
torch.cuda.set_device(0)
ctx = api.Context(0, stream=torch.cuda.current_stream(0).cuda_stream)
ops = dv.DeviceImageOps(ctx)
src = torch.empty((64, 1080, 1920), device="cuda").uniform_(0, 255)
dst = torch.empty_like(src)
for r in (5, 2):
    for _ in range(4):
        ops.gaussian(src, -1, r, out=dst)
ctx.synchronize()
