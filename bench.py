#!/usr/bin/env python3
"""bench.py -- 1080p frames/s of detect + describe + associate on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic GrayF32 frames that is already resident in HBM:
integral image -> Fast-Hessian detect -> orientation + SURF-64 (stable) describe for every frame, then greedy L2 association
(backwards validation on) of every frame with its successor in the batch.  Workload = BASELINE.json configs[1]
("Batch of 256 1920x1080 GrayF32, Fast-Hessian detect + SURF-64 describe, 1xMI355X") plus the associate leg the metric names.

Multi-GPU: frames are independent units, so each rank runs its own batch (no data-path collective, "scaling": "weak");
rank 0 prints ONE JSON line.  The line also carries
  roofline      -- the dominant kernel's algorithmic bytes / its HIP-event time measured live in the timed region
  cpu_baseline  -- the CPU oracle (C++ restatement of BoofCV's MT path) timed on a bounded sample of the same frames (rank 0, N=1)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X fp64 vector peak (spec); the exact-fp64 association path is VALU, not MFMA


def synth_frames(batch, height, width, seed0, device):
    """S-blobs(W,H,seed,n) of SURVEY 8d: background 50 + sum of n Gaussian blobs (sigma in {2,3,5,8,13,21}, amplitude +-[40,100])
    + U[0,2) noise, n ~ W*H/2000.  A blob is separable, so a frame is one [H,n]x[n,W] product."""
    n = max(1, (width * height) // 2000)
    sig_choices = torch.tensor([2.0, 3.0, 5.0, 8.0, 13.0, 21.0], device=device)
    xs = torch.arange(width, device=device, dtype=torch.float32)[None, :]
    ys = torch.arange(height, device=device, dtype=torch.float32)[:, None]
    out = torch.empty((batch, height, width), device=device, dtype=torch.float32)
    for i in range(batch):
        g = torch.Generator(device=device)
        g.manual_seed(seed0 + i)
        cx = torch.rand(n, device=device, generator=g) * width
        cy = torch.rand(n, device=device, generator=g) * height
        sig = sig_choices[torch.randint(0, 6, (n,), device=device, generator=g)]
        amp = (40.0 + 60.0 * torch.rand(n, device=device, generator=g)) * (torch.randint(0, 2, (n,), device=device, generator=g).float() * 2 - 1)
        gx = torch.exp(-((xs - cx[:, None]) ** 2) / (2 * sig[:, None] ** 2))            # [n, W]
        gy = torch.exp(-((ys - cy[None, :]) ** 2) / (2 * sig[None, :] ** 2)) * amp[None, :]  # [H, n]
        out[i] = 50.0 + gy @ gx + 2.0 * torch.rand((height, width), device=device, generator=g)
    return out


class HotPath:
    """detect + describe + associate over a device-resident batch, through the C ABI only."""

    def __init__(self, device_index, batch, height, width):
        from boofcv_amd import api, _lib
        self.api, self._lib = api, _lib
        self.L = _lib.load()
        stream = torch.cuda.current_stream(device_index).cuda_stream
        self.ctx = api.Context(device_index, stream=stream)  # kernels run on torch's current stream
        self.dd = api.FactoryDetectDescribe.surfStable(None, None, None, api.GrayF32, ctx=self.ctx)
        self.batch, self.h, self.w = batch, height, width
        self.pairs = None
        self.fit = None

    def step(self, frames):
        import ctypes as C
        B = self.batch
        self.dd.detectDevice(frames.data_ptr(), self.h * self.w, self.w, self.w, self.h, B)
        total = self.dd.totalFeatures()
        if self.pairs is None or self.pairs.numel() < total:
            self.pairs = torch.empty(max(total, 1) * 2, dtype=torch.int32, device=frames.device)
            self.fit = torch.empty(max(total, 1) * 2, dtype=torch.float64, device=frames.device)
        views = [self.dd.deviceView(i) for i in range(B)]
        counts = np.array([v[3] for v in views], dtype=np.int32)
        starts = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        base = views[0][0]  # descriptors of the whole batch are one compact [total][64] array
        # frame i (source) against frame i+1 (destination), all B problems in one batched call
        src_off = np.ascontiguousarray(starts[:B])
        dst_idx = (np.arange(B) + 1) % B
        dst_off = np.ascontiguousarray(starts[dst_idx])
        ns = np.ascontiguousarray(counts)
        nd = np.ascontiguousarray(counts[dst_idx])
        LL, I = C.POINTER(C.c_longlong), C.POINTER(C.c_int)
        st = self.L.bhip_assoc_l2_dev_batched(self.ctx._h, C.c_void_p(base), C.c_void_p(base), 64, B, src_off.ctypes.data_as(LL), ns.ctypes.data_as(I),
                                              dst_off.ctypes.data_as(LL), nd.ctypes.data_as(I), self.api.Double_MAX_VALUE, 1,
                                              C.c_void_p(self.pairs.data_ptr()), C.c_void_p(self.fit.data_ptr()))
        if st != 0:
            raise RuntimeError("bhip_assoc_l2_dev_batched failed: %s" % self.L.bhip_last_error(self.ctx._h))
        return total


def pmc_traffic(tag, batch, height, width):
    """HBM-side bytes per launch of kernel `tag` from the committed rocprofv3 PMC summary (profiles/r*_pmc_traffic.json, collected by
    scripts/pmc_traffic.sh on this workload and corrected as its header says); None when no summary matches the workload."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        wl = d.get("workload", {})
        if (wl.get("batch"), wl.get("height"), wl.get("width")) != (batch, height, width):
            continue
        name = d.get("bench_tags", {}).get(tag, tag)
        rec = d.get("kernels", {}).get(name)
        if rec:
            return rec["bytes_per_launch"], os.path.relpath(path, ROOT)
    return None, None


def cpu_baseline(frames_cpu, threads):
    """The oracle restatement of the same step on host cores: detect+describe every frame, associate consecutive frames."""
    from oracle import pyoracle as orc
    surf = orc.Surf(True)
    descs = []
    t0 = time.perf_counter()
    for f in frames_cpu:
        surf.detect(orc.Gray.from_array(f), threads=threads)
        descs.append(surf.fetch()[3].copy())
    n = len(descs)
    for i in range(n):
        orc.associate_l2(descs[i], descs[(i + 1) % n], threads=threads)
    dt = time.perf_counter() - t0
    return n / dt, dt, [len(d) for d in descs]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256, help="frames per step per GPU (BASELINE config: 256)")
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--cpu-frames", type=int, default=256, help="frames in the bounded CPU-baseline sample (0 = skip)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    B, H, W = args.batch, args.height, args.width
    frames = synth_frames(B, H, W, 1000 + rank * B, device)
    torch.cuda.synchronize()

    hp = HotPath(local_rank, B, H, W)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    kp_total = 0
    for _ in range(args.warmup):
        kp_total = hp.step(frames)
    barrier()
    hp.ctx.profile(True)
    hp.ctx.profileReset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        kp_total = hp.step(frames)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = hp.ctx.profileReport()
    hp.ctx.profile(False)

    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        k = torch.tensor([kp_total], dtype=torch.float64, device=device)
        dist.all_reduce(k, op=dist.ReduceOp.SUM)
        kp_all = float(k.item())
    else:
        kp_all = float(kp_total)

    if rank == 0:
        frames_total = world * B * args.steps
        value = frames_total / elapsed
        # dominant kernel by accumulated HIP-event time inside the timed region
        roofline = None
        if prof:
            dom = max(prof.items(), key=lambda kv: kv[1]["ms"])
            tag, r = dom
            per_launch_s = r["ms"] / r["launches"] / 1e3
            if r["bytes"] > 0:
                achieved = r["bytes"] / r["launches"] / per_launch_s / 1e9
                roofline = {"kernel": tag, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                            "avg_launch_ms": round(r["ms"] / r["launches"], 4), "launches": r["launches"]}
            else:
                # not an HBM-roofline kernel (gather + fp64 VALU per key point): report against the fp64 vector peak when flops are known
                achieved = (r["flops"] / r["launches"] / per_launch_s / 1e12) if r["flops"] > 0 else 0.0
                peak = FP32_MFMA_PEAK_TFLOPS if "mfma" in tag else FP64_VECTOR_PEAK_TFLOPS
                roofline = {"kernel": tag, "bound": "mfma", "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                            "frac": round(achieved / peak, 4), "traffic": None,
                            "avg_launch_ms": round(r["ms"] / r["launches"], 4), "launches": r["launches"]}
            tb, src = pmc_traffic(tag, B, H, W)
            if tb is not None:
                roofline["traffic"] = tb
                roofline["traffic_note"] = "bytes per launch leaving the XCD L2s (2*FETCH_SIZE+WRITE_SIZE, rocprofv3 PMC), from " + src
                roofline["algorithmic_bytes_per_launch"] = round(r["bytes"] / r["launches"]) if r["bytes"] > 0 else None
            roofline["kernels_ms_per_step"] = {k: round(v["ms"] / args.steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
            roofline["hbm_gbs_by_kernel"] = {k: round(v["bytes"] / (v["ms"] / 1e3) / 1e9, 1) for k, v in prof.items() if v["bytes"] > 0 and v["ms"] > 0}
            roofline["tflops_by_kernel"] = {k: round(v["flops"] / (v["ms"] / 1e3) / 1e12, 2) for k, v in prof.items() if v["flops"] > 0 and v["ms"] > 0}
        cpu = None
        if world == 1 and args.cpu_frames > 0:
            threads = min(os.cpu_count() or 1, 16)
            sample = frames[:args.cpu_frames].cpu().numpy()
            fps, dt, kps = cpu_baseline(sample, threads)
            cpu = {"value": round(fps, 3), "unit": "frames/s", "cores": threads, "kind": "port",
                   "sample": "%d of the same %dx%d frames, detect+describe+associate(next frame), %.1f s of CPU work, %.0f key points/frame" % (len(sample), W, H, dt, float(np.mean(kps)))}
        line = {
            "metric": "1080p frames/sec detect+describe+associate", "value": round(value, 2), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32 detect / f64 describe+associate", "data": "synthetic",
            "config": {"workload": "batch of %d %dx%d GrayF32 per GPU: Fast-Hessian detect + SURF-64 (stable) describe + greedy L2 associate with the next frame"
                                   % (B, W, H), "batch_per_gpu": B, "width": W, "height": H, "keypoints_per_frame": round(kp_all / (world * B), 1)},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
