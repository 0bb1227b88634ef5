#!/usr/bin/env python3
"""Side measurements for the BASELINE.json configs that are not the bench.py headline:
  config 3  two 4096-key-point SURF-64 sets, greedy L2 (MFMA path), 64 problems per launch
  config 4  BRIEF-512 16384 x 16384 Hamming associate on one GPU (the sharded form runs the same kernel on a row slice)
  config 5  3840x2160 frames: pyramid [1,2,4,8] (Gaussian r=2) + Sobel per layer + strict NMS on |grad|^2 is left to the caller;
            here: the pyramid and the per-layer Sobel, device resident
Prints one JSON object per part.  Inputs are resident in HBM before timing; HIP-event kernel times come from the ctx profiler."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from boofcv_amd import api, _lib  # noqa: E402


def timed(ctx, fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    ctx.synchronize()
    ctx.profile(True); ctx.profileReset()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    prof = ctx.profileReport(); ctx.profile(False)
    return dt, {k: round(v["ms"] / reps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}


def config3(ctx, L, count=64, n=4096):
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    a = torch.randn((count, n, 64), dtype=torch.float64, device="cuda", generator=g)
    a /= a.norm(dim=2, keepdim=True)
    b = a[:, torch.randperm(n, device="cuda", generator=g)] + 0.05 * torch.randn((count, n, 64), dtype=torch.float64, device="cuda", generator=g)
    b[:, 3072:] = torch.randn((count, n - 3072, 64), dtype=torch.float64, device="cuda", generator=g)
    b /= b.norm(dim=2, keepdim=True)
    pairs = torch.empty((count, n), dtype=torch.int32, device="cuda")
    fit = torch.empty((count, n), dtype=torch.float64, device="cuda")
    off = (np.arange(count, dtype=np.int64) * n)
    cnt = np.full(count, n, dtype=np.int32)
    LL, I = C.POINTER(C.c_longlong), C.POINTER(C.c_int)
    torch.cuda.synchronize()

    def run():
        st = L.bhip_assoc_l2_dev_batched(ctx._h, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), 64, count, off.ctypes.data_as(LL), cnt.ctypes.data_as(I),
                                         off.ctypes.data_as(LL), cnt.ctypes.data_as(I), api.Double_MAX_VALUE, 1, C.c_void_p(pairs.data_ptr()),
                                         C.c_void_p(fit.data_ptr()))
        assert st == 0, ctx.lastError()
    dt, k = timed(ctx, run)
    flops = 2.0 * n * n * 64 * count
    matched = int((pairs >= 0).sum().item())
    return {"part": "config3 L2 associate", "problems_per_launch": count, "n": n, "ms": round(dt * 1e3, 3), "pairs_per_s": round(count / dt, 1),
            "gemm_tflops_whole_call": round(flops / dt / 1e12, 2), "matched": matched, "kernels_ms": k}


def config4(ctx, L, n=16384, words=16):
    g = torch.Generator(device="cuda"); g.manual_seed(4)
    a = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, words), dtype=torch.int64, device="cuda", generator=g).to(torch.int32)
    flips = torch.randint(0, 65, (n,), device="cuda", generator=g)
    bitpos = torch.rand((n, 512), device="cuda", generator=g).argsort(dim=1)
    mask_bits = (torch.arange(512, device="cuda")[None, :] < flips[:, None])
    flipbits = torch.zeros((n, 512), dtype=torch.bool, device="cuda")
    flipbits.scatter_(1, bitpos, mask_bits)
    w = (flipbits.view(n, words, 32).long() << torch.arange(32, device="cuda")[None, None, :]).sum(dim=2)
    w = torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)
    b = a ^ w
    fresh = torch.randint(-2 ** 31, 2 ** 31 - 1, (n // 4, words), dtype=torch.int64, device="cuda", generator=g).to(torch.int32)
    b[-(n // 4):] = fresh
    b = b[torch.randperm(n, device="cuda", generator=g)].contiguous()
    pairs = torch.empty(n, dtype=torch.int32, device="cuda")
    fit = torch.empty(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()

    def run():
        st = L.bhip_assoc_hamming_dev(ctx._h, C.c_void_p(a.data_ptr()), n, C.c_void_p(b.data_ptr()), n, words, api.Double_MAX_VALUE, 1,
                                      C.c_void_p(pairs.data_ptr()), C.c_void_p(fit.data_ptr()))
        assert st == 0, ctx.lastError()
    dt, k = timed(ctx, run)
    return {"part": "config4 Hamming associate (1 GPU, all rows)", "n": n, "bits": words * 32, "ms": round(dt * 1e3, 3),
            "pair_scores_per_s": round(n * n / dt / 1e12, 3), "unit": "T scores/s", "matched": int((pairs >= 0).sum().item()), "kernels_ms": k}


def config5(ctx, L, batch=8, w=3840, h=2160):
    sys.path.insert(0, ROOT)
    import bench
    frames = bench.synth_frames(batch, h, w, 5000, torch.device("cuda"))
    scales = np.array([1, 2, 4, 8], np.int32)
    ker = api.FactoryKernelGaussian.gaussian1D_F32(-1, 2)
    dims = np.zeros(8, np.int32); offs = np.zeros(4, np.int64); total = C.c_longlong()
    L.bhip_pyramid_layout(w, h, scales.ctypes.data_as(_lib._ip), 4, dims.ctypes.data_as(_lib._ip), offs.ctypes.data_as(_lib._llp), C.byref(total))
    out = torch.empty((batch, total.value), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()

    def run():
        st = L.bhip_pyramid_dev_f32(ctx._h, ker.data.ctypes.data_as(_lib._fp), ker.width, scales.ctypes.data_as(_lib._ip), 4, C.c_void_p(frames.data_ptr()),
                                    w * h, w, w, h, batch, C.c_void_p(out.data_ptr()))
        assert st == 0, ctx.lastError()
    dt, k = timed(ctx, run)
    px = float(w) * h
    # algorithmic bytes per frame: layer 0 copy 8P; layer i: H pass reads prev, writes prev/skip; V pass reads that, writes prev/skip^2
    alg = 8 * px
    p = px
    for i in range(1, 4):
        alg += 4 * (p + p / 2) + 4 * (p / 2 + p / 4)
        p /= 4
    return {"part": "config5 pyramid [1,2,4,8] r=2", "batch": batch, "ms": round(dt * 1e3, 3), "frames_per_s": round(batch / dt, 1),
            "alg_GBs": round(alg * batch / dt / 1e9, 1), "kernels_ms": k}


def config2_u8(ctx, L, batch=64, w=1920, h=1080):
    """detect + describe on GrayU8 frames (host -> device upload included: the C ABI for U8 takes host buffers)."""
    import bench
    frames = bench.synth_frames(batch, h, w, 1000, torch.device("cuda")).clamp(0, 255).to(torch.uint8).cpu().numpy()
    dd = api.FactoryDetectDescribe.surfStable(None, None, None, api.GrayU8, ctx=ctx)
    imgs = [api.GrayU8.wrap(f) for f in frames]
    dt, k = timed(ctx, lambda: dd.detectBatch(imgs), reps=3, warm=1)
    return {"part": "config2 on GrayU8 frames (detect + describe, host frames)", "batch": batch, "ms": round(dt * 1e3, 2), "frames_per_s": round(batch / dt, 1),
            "keypoints_per_frame": round(dd.totalFeatures() / batch, 1), "kernels_ms": k}


def main():
    torch.cuda.set_device(0)
    ctx = api.Context(0, stream=torch.cuda.current_stream(0).cuda_stream)
    L = _lib.load()
    which = sys.argv[1:] or ["3", "4", "5"]
    for wname in which:
        r = {"3": config3, "4": config4, "5": config5, "2u8": config2_u8}[wname](ctx, L)
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()
