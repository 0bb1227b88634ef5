"""Normalised horizontal / vertical convolution with a 41-tap Gaussian (r = 20) on a resident 64 x 1080p batch: workload for rocprofv3 traces."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from boofcv_amd import api, device as dv
torch.cuda.set_device(0)
ctx = api.Context(0, stream=torch.cuda.current_stream(0).cuda_stream)
ops = dv.DeviceImageOps(ctx)
src = torch.empty((64, 1080, 1920), device="cuda").uniform_(0, 255)
dst = torch.empty_like(src)
k = api.FactoryKernelGaussian.gaussian1D_F32(-1, 20).data
for _ in range(4):
    ops.convolveNormalizedHorizontal(k, 20, src, dst)
    ops.convolveNormalizedVertical(k, 20, src, dst)
ctx.synchronize()
