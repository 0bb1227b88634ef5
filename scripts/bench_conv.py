#!/usr/bin/env python3
"""Convolution / gradient roofline (BASELINE north_star: ">= 60 % HBM roofline for convolution").

Times the device-batched boofcv-ip kernels on 64 x 1920x1080 and 8 x 3840x2160 GrayF32 batches resident in HBM:
  k_conv_h / k_conv_v (normalised Gaussian, r = 2, 5, 20), k_sobel, k_three, the pyramid's k_conv_down_h / _v (r = 2, skip 2) and the
  Gaussian blur as a whole (both passes).
Per kernel: HIP-event time per launch (ctx profiler, events on the launch stream), algorithmic bytes (8P per separable pass,
12P per gradient, 4(P_in + P_out) per down-sampling pass) and the fraction of the 8 TB/s HBM peak.  One JSON object per line;
`--json PATH` also writes the list to PATH (profiles/rNN_conv_roofline.json)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from boofcv_amd import api, device as dv  # noqa: E402

HBM_PEAK = 8000.0


def run(ctx, fn, reps):
    for _ in range(3):
        fn()
    ctx.synchronize()
    ctx.profile(True); ctx.profileReset()
    for _ in range(reps):
        fn()
    ctx.synchronize()
    prof = ctx.profileReport(); ctx.profile(False)
    return {k: (v["ms"] / v["launches"], v["bytes"] / v["launches"]) for k, v in prof.items()}


def main():
    out_path = None
    if "--json" in sys.argv:
        out_path = sys.argv[sys.argv.index("--json") + 1]
    torch.cuda.set_device(0)
    ctx = api.Context(0, stream=torch.cuda.current_stream(0).cuda_stream)
    ops = dv.DeviceImageOps(ctx)
    rows = []
    for (B, H, W) in [(64, 1080, 1920), (8, 2160, 3840)]:
        g = torch.Generator(device="cuda"); g.manual_seed(7)
        src = torch.rand((B, H, W), device="cuda", generator=g) * 255
        dst = torch.empty_like(src); dst2 = torch.empty_like(src)
        torch.cuda.synchronize()
        shape = "%dx%dx%d" % (B, W, H)

        def emit(op, prof, only=None):
            for k, (ms, nbytes) in prof.items():
                if only and k not in only:
                    continue
                gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
                r = {"op": op, "kernel": k, "batch": shape, "ms_per_launch": round(ms, 4), "algorithmic_bytes": int(nbytes), "GBs": round(gbs, 1),
                     "frac_of_hbm_peak": round(gbs / HBM_PEAK, 3)}
                rows.append(r)
                print(json.dumps(r), flush=True)

        # the device's own ceiling for one read + one write stream of this size: torch's copy kernel, timed with events on the same stream
        ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            dst.copy_(src)
        ev0.record()
        for _ in range(10):
            dst.copy_(src)
        ev1.record(); torch.cuda.synchronize()
        emit("device copy (torch, 4P read + 4P write)", {"copy": (ev0.elapsed_time(ev1) / 10, 8.0 * B * H * W)})
        for r in (2, 5, 20):
            k = api.FactoryKernelGaussian.gaussian1D_F32(-1, r).data
            emit("conv_norm_h r=%d" % r, run(ctx, lambda: ops.convolveNormalizedHorizontal(k, r, src, dst), 10))
            emit("conv_norm_v r=%d" % r, run(ctx, lambda: ops.convolveNormalizedVertical(k, r, src, dst), 10))
            p = run(ctx, lambda: ops.gaussian(src, -1, r, dst), 10)
            ms = sum(v[0] for v in p.values()); nb = sum(v[1] for v in p.values())
            emit("gaussian blur r=%d (16P algorithmic: one-pass kernel for the unrolled widths, two passes otherwise)" % r, {"+".join(sorted(p)): (ms, nb)})
        emit("sobel", run(ctx, lambda: ops.sobel(src, 0, dst, dst2), 10))
        emit("three", run(ctx, lambda: ops.three(src, 0, dst, dst2), 10))
        k2 = api.FactoryKernelGaussian.gaussian1D_F32(-1, 2).data
        p = run(ctx, lambda: ops.pyramid(k2, [1, 2, 4, 8], src), 5)
        emit("pyramid [1,2,4,8] r=2 (all layers)", p)
        del src, dst, dst2
        torch.cuda.empty_cache()
    if out_path:
        json.dump({"hbm_peak_GBs": HBM_PEAK, "rows": rows}, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
