#!/usr/bin/env python3
"""bench.py -- 1080p frames/s of detect + describe + associate on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic GrayF32 frames that is already resident in HBM:
integral image -> Fast-Hessian detect -> orientation + SURF-64 (stable) describe for every frame, then greedy L2 association
(backwards validation on) of every frame with its successor in the batch.  Workload = BASELINE.json configs[1]
("Batch of 256 1920x1080 GrayF32, Fast-Hessian detect + SURF-64 describe, 1xMI355X") plus the associate leg the metric names.

Launch: `python bench.py --gpus N` starts N rank processes itself when it is not already running under a launcher (fresh child
processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, started BEFORE this process makes any GPU call; the parent only waits and
relays rank 0's JSON line; a failing rank => non-zero exit).  Under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`
the ranks come from the environment; WORLD_SIZE != --gpus is an error, never a silent single-GPU run.

Workloads (--workload):
  frames         (default) the headline metric.  Frames are independent units: each rank runs its own batch, no data-path collective,
                 "scaling": "weak".  Rank 0 prints ONE JSON line, which also carries
                   roofline      the dominant kernel's algorithmic bytes / its HIP-event time measured live in the timed region
                   cpu_baseline  the CPU oracle (C++ restatement of BoofCV's MT path) on a bounded sample of the same frames (N=1 only)
                   end_to_end    the same step through the host-buffer boundary a JNI caller uses (bhip_surf_detect_f32 -> bhip_surf_fetch ->
                                 bhip_assoc_l2_f64): host frames in, descriptors and matches out, PCIe included (never `value`)
  assoc_sharded  BASELINE configs[3]: one 16384 x 16384 BRIEF-512 Hamming association sharded over the ranks (boofcv_amd/sharded.py:
                 local phase 1 -> ONE all_gather_into_tensor of the column top-2 records over RCCL -> local phase 2), "scaling": "strong"
  chain4k        BASELINE configs[4]: 3840x2160 frames through pyramid -> Sobel -> |grad|^2 -> strict NMS per layer + Fast-Hessian/SURF on
                 layer 0, device resident, frames sharded over the ranks ("weak")
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
CU_COUNT = 256                # MI355X: 8 XCDs x 32 CUs
PROFILE_EVERY = 4             # per-kernel HIP events on every 4th step of the timed region (see run_frames)
ENGINE_CLOCK_HZ = 2.4e9       # peak engine clock (MI355X_MICROARCH.md)
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
FP16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense fp16 / bf16 matrix peak (v_mfma_f32_32x32x16_f16: what k_assoc_mfma_* issue)
INT8_MFMA_PEAK_TOPS = 5000.0  # MI355X_MICROARCH.md: I8 MFMA = 2x the BF16 rate (~2.5 PF dense)
FP64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X fp64 vector peak (spec); the exact-fp64 association path is VALU, not MFMA
PCIE_GBS = 63.0  # MI355X_MICROARCH.md: host link PCIe Gen5 x16 (spec)
_T0 = time.perf_counter()
MIN_TIMED_SECONDS = 3.0  # default --steps: enough steps for a timed region an external GPU-busy sampler can see


# ---------------------------------------------------------------------------------------------------------------- launcher
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: as many as fill %.0f s, at least 3)" % MIN_TIMED_SECONDS)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=["frames", "assoc_sharded", "chain4k", "brief_frames"], default="frames")
    ap.add_argument("--batch", type=int, default=None, help="frames per step per GPU (frames: 256 = BASELINE config; chain4k: 8)")
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--cpu-frames", type=int, default=256, help="frames in the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host-boundary (PCIe-inclusive) leg")
    ap.add_argument("--end-to-end", action="store_true", help="run the host-boundary leg at --gpus > 1 too (default there: skipped, the scaling run stays short)")
    ap.add_argument("--no-conv", action="store_true", help="skip the convolution front-end leg (Gaussian blur, Sobel, pyramid layer on 64 of the resident frames)")
    ap.add_argument("--dry-launch", action="store_true", help="ranks only report their environment; no GPU call anywhere (launcher test)")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(args, argv):
    """Parent of a self-launched multi-GPU run.  Makes no GPU call (torch is not even imported here): starts one child per rank,
    relays rank 0's stdout (the JSON line), forwards the other ranks' output to stderr, exits non-zero if any rank failed."""
    n = args.gpus
    env0 = dict(os.environ)
    env0.setdefault("MASTER_ADDR", "127.0.0.1")
    env0["MASTER_PORT"] = str(free_port())
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [None] * n

    def drain(i):
        outs[i] = procs[i].communicate()

    threads = [threading.Thread(target=drain, args=(i,)) for i in range(n)]
    for t in threads:
        t.start()
    # if one rank dies the others would wait in a collective for ever: watch and terminate the exact processes we started
    failed = False
    while any(t.is_alive() for t in threads):
        for p in procs:
            rc = p.poll()
            if rc is not None and rc != 0 and not failed:
                failed = True
                for q in procs:
                    if q.poll() is None:
                        q.terminate()
        time.sleep(0.2)
    for t in threads:
        t.join()
    rc = 0
    for r, p in enumerate(procs):
        out, err = outs[r]
        if err:
            sys.stderr.write("".join("[rank %d] %s\n" % (r, line) for line in err.splitlines()))
        if r == 0:
            # the contract is ONE JSON line on stdout: anything else a library printed there (gloo's connection banner in rehearsal mode) goes to stderr
            for line in out.splitlines():
                (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line + "\n")
        elif out:
            sys.stderr.write("".join("[rank %d] %s\n" % (r, line) for line in out.splitlines()))
        if p.returncode != 0:
            rc = p.returncode or 1
    sys.stdout.flush()
    return rc


# ---------------------------------------------------------------------------------------------------------------- synthetic inputs
def synth_frames(batch, height, width, seed0, device):
    """S-blobs(W,H,seed,n) of SURVEY 8d: background 50 + sum of n Gaussian blobs (sigma in {2,3,5,8,13,21}, amplitude +-[40,100])
    + U[0,2) noise, n ~ W*H/2000.  A blob is separable, so a frame is one [H,n]x[n,W] product."""
    import torch
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


def synth_brief(n, words, seed, device):
    """Config 4 inputs: A = random int32 words; B = A with k in 0..64 random bit flips for 75 % of the rows, fresh rows for 25 %,
    rows shuffled.  Every rank builds the same sets from the same seed."""
    import torch
    g = torch.Generator(device=device); g.manual_seed(seed)
    a = torch.randint(-2 ** 31, 2 ** 31 - 1, (n, words), dtype=torch.int64, device=device, generator=g).to(torch.int32)
    flips = torch.randint(0, 65, (n,), device=device, generator=g)
    bitpos = torch.rand((n, words * 32), device=device, generator=g).argsort(dim=1)
    mask_bits = (torch.arange(words * 32, device=device)[None, :] < flips[:, None])
    flipbits = torch.zeros((n, words * 32), dtype=torch.bool, device=device)
    flipbits.scatter_(1, bitpos, mask_bits)
    w = (flipbits.view(n, words, 32).long() << torch.arange(32, device=device)[None, None, :]).sum(dim=2)
    w = torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32)
    b = a ^ w
    b[-(n // 4):] = torch.randint(-2 ** 31, 2 ** 31 - 1, (n // 4, words), dtype=torch.int64, device=device, generator=g).to(torch.int32)
    b = b[torch.randperm(n, device=device, generator=g)].contiguous()
    return a, b


# ---------------------------------------------------------------------------------------------------------------- the hot path
class HotPath:
    """detect + describe + associate over a device-resident batch, through the C ABI only."""

    def __init__(self, device_index, batch, height, width):
        import torch
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
        import numpy as np
        import torch
        B = self.batch
        self.dd.detectDevice(frames.data_ptr(), self.h * self.w, self.w, self.w, self.h, B)
        total = self.dd.totalFeatures()
        if self.pairs is None or self.pairs.numel() < total:
            self.pairs = torch.empty(max(total, 1) * 2, dtype=torch.int32, device=frames.device)
            self.fit = torch.empty(max(total, 1) * 2, dtype=torch.float64, device=frames.device)
        counts = self.dd.counts()                 # one native call (256 per-image calls cost the host ~1 ms with the GPU idle)
        starts = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        base = self.dd.deviceView(0)[0]           # descriptors of the whole batch are one compact [total][64] array
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


class HostBoundary:
    """The same step through the host-buffer entry points a JNI shim binds: frames in pinned host memory ->
    bhip_surf_detect_f32 (sub-batches) -> bhip_surf_fetch per frame (location, orientation, sign, descriptor to host) ->
    bhip_assoc_l2_f64 per consecutive pair (host descriptors in, matches out).  Two host threads, each with its own ctx / stream /
    detector, take alternate sub-batches so the upload of one overlaps the kernels of the other."""

    def __init__(self, device_index, frames_host, sub_batch):
        import numpy as np
        from boofcv_amd import api
        self.api = api
        self.np = np
        self.dev = device_index
        self.frames = frames_host      # pinned [B,H,W] float32 numpy view
        self.B, self.h, self.w = frames_host.shape
        self.sub = sub_batch
        self.workers = []
        for _ in range(2):
            ctx = api.Context(device_index)
            dd = api.FactoryDetectDescribe.surfStable(None, None, None, api.GrayF32, ctx=ctx)
            assoc = api.FactoryAssociation.greedy(api.ScoreAssociateEuclideanSq_F64(), api.Double_MAX_VALUE, True, ctx=ctx)
            self.workers.append((ctx, dd, assoc))
        self.desc = [None] * self.B
        self.matches = 0

    _out = None

    def step_batched(self, frames_u8=None):
        """The same work through the library's batch-level calls: one bhip_surf_detect_f32 over the whole batch of pinned host frames, one
        bhip_surf_fetch_all (every location / orientation / sign / descriptor to host), one bhip_assoc_l2_surf over the descriptors still
        resident from that detect (matches to host).  What a provider's detectBatch / associate pair costs when it avoids per-frame calls."""
        api = self.api
        ctx, dd, _ = self.workers[0]
        if frames_u8 is not None:   # GrayU8 frames: a quarter of the ingest bytes (GrayS32 integral images on the device)
            imgs = [api.GrayU8(self.w, self.h, frames_u8[i].reshape(-1)) for i in range(self.B)]
        else:
            imgs = [api.GrayF32(self.w, self.h, self.frames[i].reshape(-1)) for i in range(self.B)]
        dd.detectBatch(imgs)
        total = dd.totalFeatures()
        if self._out is None or self._out[1].shape[0] < total:
            # page-locked result arrays, kept across batches (a provider's reusable result store)
            import torch
            cap = int(total * 1.25) + 1024
            self._pins = [torch.empty((cap, 3), dtype=torch.float64, pin_memory=True), torch.empty(cap, dtype=torch.float64, pin_memory=True),
                          torch.empty(cap, dtype=torch.uint8, pin_memory=True), torch.empty((cap, 64), dtype=torch.float64, pin_memory=True)]
            self._out = tuple(t.numpy() for t in self._pins)
        xys, ang, white, desc, starts = dd.fetchAll(out=self._out)
        src = self.np.arange(self.B, dtype=self.np.int32)
        pairs, fit = dd.associateImages(src, (src + 1) % self.B)
        self.matches = int((pairs[:int(starts[-1])] >= 0).sum())
        return int(starts[-1])

    def step(self):
        """The reference's per-call interface as a two-stage pipeline: one host thread detects (sub-batches) and fetches frame by frame, a
        second one associates every consecutive pair as soon as both descriptor lists have arrived -- detect is GPU-bound, the per-pair
        association is bound by its two pageable descriptor uploads, so the two stages overlap (round 2 ran two detect threads, then two
        association threads: slower than ONE thread, scripts/e2e_strict_breakdown.py)."""
        chunks = [(a, min(a + self.sub, self.B)) for a in range(0, self.B, self.sub)]
        ready = [threading.Event() for _ in range(self.B)]
        self.desc = [None] * self.B
        err = []

        def detect_stage():
            try:
                api = self.api
                ctx, dd, _ = self.workers[0]
                for (a, b) in chunks:
                    imgs = [api.GrayF32(self.w, self.h, self.frames[i].reshape(-1)) for i in range(a, b)]
                    dd.detectBatch(imgs)
                    for j in range(b - a):
                        self.desc[a + j] = dd._results(j)[3]    # bhip_surf_fetch: xy/scale, angle, sign, descriptors to host
                        ready[a + j].set()
            except Exception as e:   # noqa: BLE001 -- reported by the caller
                err.append(e)
                for ev in ready:
                    ev.set()

        def assoc_stage():
            try:
                _, _, assoc = self.workers[1]
                m = 0
                for i in range(self.B):
                    j = (i + 1) % self.B
                    ready[i].wait(); ready[j].wait()
                    if err:
                        return
                    assoc.setSource(self.desc[i]); assoc.setDestination(self.desc[j]); assoc.associate()
                    m += int((assoc.getPairs() >= 0).sum())
                self.matches = m
            except Exception as e:   # noqa: BLE001
                err.append(e)

        ts = [threading.Thread(target=detect_stage), threading.Thread(target=assoc_stage)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if err:
            raise err[0]
        return sum(len(d) for d in self.desc)


def trace(msg):
    """progress line on stderr when BHIP_BENCH_TRACE is set (profiler runs: shows which stage a slow counter pass is in)"""
    if os.environ.get("BHIP_BENCH_TRACE"):
        sys.stderr.write("[bench %.1fs] %s\n" % (time.perf_counter() - _T0, msg))
        sys.stderr.flush()


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


class Dist:
    """rank plumbing: torch.distributed over RCCL ("nccl") when WORLD_SIZE > 1"""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        # Rehearsal on a box with fewer GPUs than ranks (BHIP_BENCH_REHEARSAL=1, never a quoted number): the ranks share the visible
        # devices and the control collectives (barrier, scalar all-reduce) go over gloo -- RCCL cannot put two ranks on one device.
        self.rehearsal = os.environ.get("BHIP_BENCH_REHEARSAL") == "1"
        dev_index = self.local_rank % max(torch.cuda.device_count(), 1) if self.rehearsal else self.local_rank
        self.ctl_device = torch.device("cpu") if self.rehearsal else torch.device("cuda", dev_index)
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if self.rehearsal:
                dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group(backend="nccl", rank=self.rank, world_size=self.world, device_id=torch.device("cuda", dev_index))
            self.dist = dist
        torch.cuda.set_device(dev_index)
        self.local_rank = dev_index
        self.device = torch.device("cuda", dev_index)

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max(self, x):
        if self.dist is None:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.ctl_device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, x):
        if self.dist is None:
            return x
        t = self.torch.tensor([x], dtype=self.torch.float64, device=self.ctl_device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def ranks_seen(self):
        return self.dist.get_world_size() if self.dist is not None else 1

    def device_ids(self):
        """the device index every rank runs on, in rank order (one all-gather of one integer per rank)"""
        if self.dist is None:
            return [self.local_rank]
        mine = self.torch.tensor([self.local_rank], dtype=self.torch.int64, device=self.ctl_device)
        out = [self.torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        return [int(t.item()) for t in out]

    def close(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def timed_steps(D, args, step_fn):
    """W untimed warm-up steps, then exactly K steps between barrier + synchronize on both sides; elapsed = MAX over ranks.
    K = --steps, or (when not given) the number of steps that fills MIN_TIMED_SECONDS, agreed across ranks."""
    torch = D.torch
    result = None
    for _ in range(max(args.warmup, 0)):
        result = step_fn()
    steps = args.steps
    if steps is None:
        D.barrier()
        t0 = time.perf_counter()
        result = step_fn()
        torch.cuda.synchronize()
        one = D.max(time.perf_counter() - t0)
        steps = max(3, int(MIN_TIMED_SECONDS / max(one, 1e-6)) + 1)
        steps = int(D.max(float(steps)))
    D.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        result = step_fn()
    torch.cuda.synchronize()
    D.barrier()
    elapsed = D.max(time.perf_counter() - t0)
    return steps, elapsed, result


# ---------------------------------------------------------------------------------------------------------------- workload: frames
def run_frames(args, D):
    import numpy as np
    torch = D.torch
    B = args.batch or 256
    H = args.height or 1080
    W = args.width or 1920
    # profiler runs (scripts/pmc_*.sh): the synthesis (256 GEMMs) is done once outside the profiler and re-read from a scratch file --
    # rocprofv3 counter collection has hung in torch's GEMM kernels on this pool
    cache = os.environ.get("BHIP_BENCH_FRAMES_CACHE")
    cache = "%s.%d_%dx%dx%d" % (cache, D.rank, B, H, W) if cache else None
    if cache and os.path.exists(cache):
        trace("reading %d frames from %s" % (B, cache))
        frames = torch.from_numpy(np.fromfile(cache, dtype=np.float32).reshape(B, H, W)).to(D.device)
    else:
        trace("synthesising %d frames" % B)
        frames = synth_frames(B, H, W, 1000 + D.rank * B, D.device)
        if cache:
            frames.cpu().numpy().tofile(cache)
    torch.cuda.synchronize()
    trace("frames resident")
    hp = HotPath(D.local_rank, B, H, W)

    kp = [0]
    # Per-kernel HIP events (the live roofline numbers) bracket every launch of every PROFILE_EVERY-th step of the timed region: an event
    # pair costs ~10 us of dependency latency per kernel, ~0.4 ms over the ~40 kernels of a step, so bracketing every step would take 2 %
    # off the throughput it measures.  The sampled steps are ordinary steps of the timed region.
    sampling = {"on": False, "n": 0, "sampled": 0}

    def step():
        on = sampling["on"] and sampling["n"] % PROFILE_EVERY == 0
        hp.ctx.profile(on)
        sampling["n"] += 1
        sampling["sampled"] += 1 if on else 0
        kp[0] = hp.step(frames)
        trace("step done: %d key points" % kp[0])
        return kp[0]

    for _ in range(max(args.warmup, 0)):
        step()
    warm = args.warmup
    args.warmup = 0
    if args.steps is None:
        D.barrier()
        t0 = time.perf_counter(); step(); torch.cuda.synchronize()
        one = D.max(time.perf_counter() - t0)
        args.steps = int(D.max(float(max(3, int(MIN_TIMED_SECONDS / max(one, 1e-6)) + 1))))
    sampling.update(on=True, n=0, sampled=0)
    hp.ctx.profileReset()
    steps, elapsed, _ = timed_steps(D, args, step)
    args.warmup = warm
    prof = hp.ctx.profileReport()
    hp.ctx.profile(False)
    psteps = max(sampling["sampled"], 1)   # steps whose kernels were bracketed
    kp_all = D.sum(float(kp[0]))

    # host-boundary leg (every rank runs it at the same time: the ranks share the host's PCIe complex and cores)
    e2e = None
    if not args.no_end_to_end and (D.world == 1 or args.end_to_end):
        host = torch.empty((B, H, W), dtype=torch.float32, pin_memory=True)
        host.copy_(frames)
        torch.cuda.synchronize()
        hb = HostBoundary(D.local_rank, host.numpy(), sub_batch=min(int(os.environ.get("BHIP_BENCH_SUB_BATCH", "64")), B))
        hb.step()   # warm-up: allocations, first-touch
        D.barrier()
        reps = 2 if B >= 64 else 5
        t0 = time.perf_counter()
        for _ in range(reps):
            kp_e2e = hb.step()
        D.barrier()
        dt = D.max(time.perf_counter() - t0)
        matches_strict = hb.matches
        desc_bytes = kp_e2e * (64 * 8 + 3 * 8 + 8 + 1)
        h2d = B * H * W * 4 + 2 * kp_e2e * 64 * 8          # frames + both descriptor sets of every association
        d2h = desc_bytes + kp_e2e * 12
        # batch-level calls of the same library (detectBatch + fetch_all + resident association)
        hb.step_batched()
        D.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            kp_b = hb.step_batched()
        D.barrier()
        dtb = D.max(time.perf_counter() - t0)
        # the same batch-level calls on GrayU8 frames (the frames rounded to bytes, page-locked): 2.07 MB instead of 8.29 MB per frame over PCIe
        host_u8 = torch.empty((B, H, W), dtype=torch.uint8, pin_memory=True)
        host_u8.copy_(frames.clamp(0, 255).to(torch.uint8))
        torch.cuda.synchronize()
        u8 = host_u8.numpy()
        hb.step_batched(u8)
        D.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            kp_u8 = hb.step_batched(u8)
        D.barrier()
        dtu = D.max(time.perf_counter() - t0)
        batched_u8 = {"value": round(D.world * B * reps / dtu, 1), "unit": "frames/s", "ms_per_batch": round(1e3 * dtu / reps, 2),
                      "path": "the same three calls on GrayU8 frames (bhip_surf_detect_u8: GrayS32 integral images, integer taps)",
                      "h2d_bytes_per_frame": H * W, "d2h_bytes_per_frame": int((kp_u8 * (64 * 8 + 3 * 8 + 8 + 1) + kp_u8 * 12) / B),
                      "pcie_ceiling_frames_per_s": round(PCIE_GBS * 1e9 / (H * W), 1), "keypoints_per_frame": round(kp_u8 / B, 1)}
        del host_u8
        batched = {"value": round(D.world * B * reps / dtb, 1), "unit": "frames/s", "ms_per_batch": round(1e3 * dtb / reps, 2), "gray_u8": batched_u8,
                   "path": "one bhip_surf_detect_f32 over the %d pinned host frames + one bhip_surf_fetch_all into page-locked result arrays + one bhip_assoc_l2_surf (descriptors stay "
                           "resident for the association, matches come back to the host)" % B,
                   "h2d_bytes_per_frame": H * W * 4, "d2h_bytes_per_frame": int((kp_b * (64 * 8 + 3 * 8 + 8 + 1) + kp_b * 12) / B),
                   "pcie_ceiling_frames_per_s": round(PCIE_GBS * 1e9 / (H * W * 4), 1), "matches_per_frame": round(hb.matches / B, 1)}
        e2e = {"value": round(D.world * B * reps / dt, 1), "unit": "frames/s", "ms_per_batch": round(1e3 * dt / reps, 2),
               "path": "bhip_surf_detect_f32 (pinned host frames, sub-batches of %d: chunks of 32 whose upload runs under the previous chunk's kernels) + "
                       "bhip_surf_fetch per frame into page-locked arrays on one host thread, bhip_assoc_l2_f64 per consecutive pair on a second one as the "
                       "descriptor lists arrive" % hb.sub,
               "h2d_bytes_per_frame": int(h2d / B), "d2h_bytes_per_frame": int(d2h / B),
               "pcie_ceiling_frames_per_s": round(PCIE_GBS * 1e9 / (h2d / B), 1), "pcie_peak_GBs": PCIE_GBS,
               "matches_per_frame": round(matches_strict / B, 1), "batched_calls": batched}
        del hb

    # convolution front end on the same resident frames (north_star: ">= 60 % HBM roofline for convolution"): one short leg so that the
    # driver's line carries it -- Gaussian blur r = 2 and r = 5 (one-pass kernel; 16P algorithmic bytes = the two separable passes of
    # SURVEY 8d), Sobel (12P), the discrete pyramid's first down-sampling layer (4(P + P/2) + 4(P/2 + P/4))
    conv = None
    devices = D.device_ids()
    if D.rank == 0 and not args.no_conv and D.world == 1:
        from boofcv_amd import device as dv
        ops = dv.DeviceImageOps(hp.ctx)
        nb = min(B, 64)
        src = frames[:nb]
        dst = torch.empty_like(src); dst2 = torch.empty_like(src)
        ker2 = hp.api.FactoryKernelGaussian.gaussian1D_F32(-1, 2).data
        P = float(nb * H * W)
        legs = [("gaussian_r2", lambda: ops.gaussian(src, -1, 2, dst), 16 * P), ("gaussian_r5", lambda: ops.gaussian(src, -1, 5, dst), 16 * P),
                ("sobel", lambda: ops.sobel(src, 0, dst, dst2), 12 * P), ("pyramid_1_2", lambda: ops.pyramid(ker2, [1, 2], src), None)]
        conv = {"batch": "%dx%dx%d" % (nb, W, H), "peak_GBs": HBM_PEAK_GBS}
        for name, fn, alg in legs:
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            hp.ctx.profile(True); hp.ctx.profileReset()
            reps = 5
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            pr = hp.ctx.profileReport(); hp.ctx.profile(False)
            ms = sum(v["ms"] for v in pr.values()) / reps
            nbytes = alg if alg is not None else sum(v["bytes"] for v in pr.values()) / reps
            gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            conv[name] = {"ms": round(ms, 4), "GBs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 3), "kernels": sorted(pr)}
        del dst, dst2
    if D.rank != 0:
        return None
    frames_total = D.world * B * steps
    value = frames_total / elapsed
    roofline = None
    if prof:
        tag, r = max(prof.items(), key=lambda kv: kv[1]["ms"])
        per_launch_s = r["ms"] / r["launches"] / 1e3
        if r["bytes"] > 0:
            achieved = r["bytes"] / r["launches"] / per_launch_s / 1e9
            roofline = {"kernel": tag, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                        "avg_launch_ms": round(r["ms"] / r["launches"], 4), "launches": r["launches"],
                        "algorithmic_bytes_per_launch": round(r["bytes"] / r["launches"])}
        else:
            achieved = (r["flops"] / r["launches"] / per_launch_s / 1e12) if r["flops"] > 0 else 0.0
            # the association's matrix-core passes run v_mfma_f32_32x32x16_f16 (fp16 candidate filter); everything else with flops is fp64 VALU
            peak = FP16_MFMA_PEAK_TFLOPS if tag.startswith("k_assoc_mfma") else FP32_MFMA_PEAK_TFLOPS if "mfma" in tag else FP64_VECTOR_PEAK_TFLOPS
            roofline = {"kernel": tag, "bound": "mfma", "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                        "frac": round(achieved / peak, 4), "traffic": None,
                        "avg_launch_ms": round(r["ms"] / r["launches"], 4), "launches": r["launches"]}
        tb, src = pmc_traffic(tag, B, H, W)
        if tb is not None:
            roofline["traffic"] = tb
            roofline["traffic_note"] = "bytes per launch leaving the XCD L2s (2*FETCH_SIZE+WRITE_SIZE, rocprofv3 PMC), from " + src
            roofline["hbm_frac_from_traffic"] = round(tb / per_launch_s / 1e9 / HBM_PEAK_GBS, 4)
        if tag == "k_describe":
            # a gather stage.  `achieved` follows the contract (algorithmic tap bytes / time) but nearly all of those bytes are
            # served by L1/L2.  What bounds the kernel is the load path's request rate: a wave-wide gather costs one cycle per lane request
            # (scripts/probe/line_cost.hip), the kernel issues 8146 lane requests per key point (TCP_TOTAL_CACHE_ACCESSES, profiles/r03_pmc_ta.txt:
            # 12 taps = 10 requests for each of the 289 + 576 samples) -- DESIGN.md section 4, "the load path's cost rule".
            kps_per_launch = r["bytes"] / r["launches"] / 42041.0   # the per-key-point byte figure the launch was priced with (stable defaults)
            req = 8146.0 * kps_per_launch
            rate = req / (per_launch_s * CU_COUNT * ENGINE_CLOCK_HZ)
            roofline["limiter"] = "load-path request rate (one cycle per lane request and CU; cache-served gathers -- see hbm_frac_from_traffic for the bytes that reach HBM)"
            roofline["load_path"] = {"lane_requests_per_key_point": 8146, "key_points_per_launch": round(kps_per_launch), "requests_per_cycle_per_cu": round(rate, 3),
                                     "peak_requests_per_cycle_per_cu": 1.0, "frac": round(rate, 3),
                                     "note": "request count from rocprofv3 PMC (profiles/r03_pmc_ta.txt), rate rule from scripts/probe/line_cost.hip (profiles/r03_quad_gather_experiment.txt)"}
        roofline["kernels_ms_per_step"] = {k: round(v["ms"] / psteps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}
        roofline["steps_with_kernel_events"] = "%d of %d (every %d. step of the timed region)" % (psteps, steps, PROFILE_EVERY)
        roofline["hbm_gbs_by_kernel"] = {k: round(v["bytes"] / (v["ms"] / 1e3) / 1e9, 1) for k, v in prof.items() if v["bytes"] > 0 and v["ms"] > 0}
        roofline["tflops_by_kernel"] = {k: round(v["flops"] / (v["ms"] / 1e3) / 1e12, 2) for k, v in prof.items() if v["flops"] > 0 and v["ms"] > 0}
        # the association stage as a whole: the N x M x 64 contraction of SURVEY 8d counted ONCE (the two matrix-core passes each issue it)
        as_ms = sum(v["ms"] for k, v in prof.items() if k.startswith("k_assoc"))
        p1 = prof.get("k_assoc_mfma_pass1")
        if as_ms > 0 and p1 and p1["flops"] > 0:
            tf = p1["flops"] / (as_ms / 1e3) / 1e12
            roofline["assoc_stage"] = {"ms_per_step": round(as_ms / psteps, 3), "algorithmic_TFLOPs": round(tf, 1),
                                       "stage_frac_fp16_mfma_peak": round(tf / FP16_MFMA_PEAK_TFLOPS, 4), "stage_frac_fp32_mfma_peak": round(tf / FP32_MFMA_PEAK_TFLOPS, 4),
                                       "note": "fp16 MFMA candidate filter (two passes) + exact fp64 re-score of the listed pairs; results are exact"}
        # the detect stage as a whole (a1-a4) against its HBM roofline: SURVEY 8d's 66.5 * P bytes per frame
        det_ms = sum(v["ms"] for k, v in prof.items() if k.startswith(("k_integral", "k_detect_fused", "k_hessian", "k_nms", "k_word_prefix", "k_rank_scatter")))
        if det_ms > 0:
            det_gbs = 66.5 * W * H * B * psteps / (det_ms / 1e3) / 1e9
            roofline["detect_stage"] = {"ms_per_step": round(det_ms / psteps, 3), "algorithmic_GBs": round(det_gbs, 1), "frac": round(det_gbs / HBM_PEAK_GBS, 4)}
    cpu = None
    if D.world == 1 and args.cpu_frames > 0:
        threads = min(os.cpu_count() or 1, 16)
        sample = frames[:args.cpu_frames].cpu().numpy()
        fps, dt, kps = cpu_baseline(sample, threads)
        cpu = {"value": round(fps, 3), "unit": "frames/s", "cores": threads, "kind": "port",
               "sample": "%d of the same %dx%d frames, detect+describe+associate(next frame), %.1f s of CPU work, %.0f key points/frame" % (len(sample), W, H, dt, float(np.mean(kps)))}
    return {
        "metric": "1080p frames/sec detect+describe+associate", "value": round(value, 2), "unit": "frames/s", "n_gpus": D.world,
        "steps": steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / steps, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 detect / f64 describe / associate: fp16 MFMA candidate filter + exact fp64 re-score", "data": "synthetic",
        "config": {"workload": "batch of %d %dx%d GrayF32 per GPU: Fast-Hessian detect + SURF-64 (stable) describe + greedy L2 associate with the next frame"
                               % (B, W, H), "batch_per_gpu": B, "width": W, "height": H, "keypoints_per_frame": round(kp_all / (D.world * B), 1),
                   "ranks_seen": D.ranks_seen(), "device_ids": devices},
        "roofline": roofline, "cpu_baseline": cpu, "end_to_end": e2e, "conv": conv,
    }


# ---------------------------------------------------------------------------------------------------------------- workload: assoc_sharded
def run_assoc_sharded(args, D):
    torch = D.torch
    from boofcv_amd import sharded, api
    n, words = 16384, 16
    a, b = synth_brief(n, words, 4, D.device)
    part = sharded.row_partition(n, D.world)
    begin, count = part[D.rank]
    src_local = a[begin:begin + count].contiguous()
    eng = sharded.GpuEngine(device=D.local_rank)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    gather_ms = [0.0]
    res = [None]

    def step():
        # the same orchestration as sharded.associate_sharded, with events around the one exchange step
        pairs, fit, col = eng.phase1("hamming", src_local, begin, b, api.Double_MAX_VALUE)
        if D.world > 1:
            col_all = torch.empty(D.world * col.numel(), dtype=torch.uint8, device=col.device)
            ev[0].record()
            D.dist.all_gather_into_tensor(col_all, col.contiguous())
            ev[1].record()
        else:
            col_all = col
        res[0] = eng.phase2(col_all, D.world, n, pairs, fit, begin)
        if D.world > 1:
            ev[1].synchronize()
            gather_ms[0] += ev[0].elapsed_time(ev[1])
        return res[0]

    for _ in range(max(args.warmup, 0)):
        step()
    warm = args.warmup
    args.warmup = 0
    gather_ms[0] = 0.0
    eng.ctx.profile(True); eng.ctx.profileReset()
    if args.steps is None:
        D.barrier()
        t0 = time.perf_counter(); step(); torch.cuda.synchronize()
        one = D.max(time.perf_counter() - t0)
        args.steps = int(D.max(float(max(3, int(MIN_TIMED_SECONDS / max(one, 1e-6)) + 1))))
        gather_ms[0] = 0.0
        eng.ctx.profileReset()
    steps, elapsed, _ = timed_steps(D, args, step)
    args.warmup = warm
    prof = eng.ctx.profileReport(); eng.ctx.profile(False)
    matched = D.sum(float((res[0][0] >= 0).sum().item()))
    gather = D.max(gather_ms[0] / steps)
    if D.rank != 0:
        return None
    ms = 1e3 * elapsed / steps
    dom = max(prof.items(), key=lambda kv: kv[1]["ms"]) if prof else None
    roofline = None
    if dom:
        tag, r = dom
        per_launch_s = r["ms"] / r["launches"] / 1e3
        tops = r["flops"] / r["launches"] / per_launch_s / 1e12 if r["flops"] > 0 else 0.0
        roofline = {"kernel": tag, "bound": "mfma", "achieved": round(tops, 2), "peak": INT8_MFMA_PEAK_TOPS, "unit": "TOP/s (int8)",
                    "frac": round(tops / INT8_MFMA_PEAK_TOPS, 4), "traffic": None, "avg_launch_ms": round(r["ms"] / r["launches"], 4), "launches": r["launches"],
                    "kernels_ms_per_step": {k: round(v["ms"] / steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}}
    return {
        "metric": "BRIEF-512 16384x16384 Hamming associations/s (sharded over the ranks)", "value": round(steps / elapsed, 2), "unit": "associations/s",
        "n_gpus": D.world, "steps": steps, "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "u32 popcount as int8 MFMA", "data": "synthetic",
        "config": {"workload": "one 16384x16384 BRIEF-512 greedy Hamming association with backwards validation; rank r owns %d source rows and the whole "
                               "destination set; one all_gather_into_tensor of 16384 24-byte column records per rank" % part[0][1],
                   "ranks_seen": D.ranks_seen(), "all_gather_ms": round(gather, 4), "all_gather_share": round(gather / ms, 4) if ms > 0 else None,
                   "matched": int(matched)},
        "roofline": roofline, "cpu_baseline": None,
    }


# ---------------------------------------------------------------------------------------------------------------- workload: chain4k
def run_chain4k(args, D):
    import numpy as np
    torch = D.torch
    from boofcv_amd import api, device as dv
    B = args.batch or 8
    H = args.height or 2160
    W = args.width or 3840
    frames = synth_frames(B, H, W, 5000 + D.rank * B, D.device)
    ctx = api.Context(D.local_rank, stream=torch.cuda.current_stream(D.local_rank).cuda_stream)
    ops = dv.DeviceImageOps(ctx)
    dd = api.FactoryDetectDescribe.surfStable(None, None, None, api.GrayF32, ctx=ctx)
    ker = api.FactoryKernelGaussian.gaussian1D_F32(-1, 2).data
    scales = [1, 2, 4, 8]
    dims, offs, total = ops.pyramidLayout(W, H, scales)
    # every buffer of the chain is allocated once: steps reuse them, as a streaming caller would
    bufs = {"dx": [], "dy": [], "sq": []}
    for (w, h) in dims:
        for k in bufs:
            bufs[k].append(torch.empty((B, int(h), int(w)), dtype=torch.float32, device=D.device))
    counts = [0, 0]

    def step():
        layers = ops.pyramid(ker, scales, frames)
        nms = 0
        ns = []
        for i, layer in enumerate(layers):
            ops.sobel(layer, 0, bufs["dx"][i], bufs["dy"][i])
            ops.intensity(dv.INTENSITY_SQ, bufs["dx"][i], bufs["dy"][i], bufs["sq"][i])
            xy, n = ops.nonmax(bufs["sq"][i], 2, 25.0, 2)
            ns.append(n)
        l0 = layers[0]
        dd.detectDevice(l0.data_ptr(), l0.stride(0), l0.stride(1), W, H, B)
        counts[0] = dd.totalFeatures()
        counts[1] = int(torch.stack([n.sum() for n in ns]).sum().item())
        return counts[0]

    for _ in range(max(args.warmup, 0)):
        step()
    warm = args.warmup
    args.warmup = 0
    ctx.profile(True); ctx.profileReset()
    if args.steps is None:
        D.barrier()
        t0 = time.perf_counter(); step(); torch.cuda.synchronize()
        one = D.max(time.perf_counter() - t0)
        args.steps = int(D.max(float(max(3, int(MIN_TIMED_SECONDS / max(one, 1e-6)) + 1))))
        ctx.profileReset()
    steps, elapsed, _ = timed_steps(D, args, step)
    args.warmup = warm
    prof = ctx.profileReport(); ctx.profile(False)
    kp_all = D.sum(float(counts[0]))
    nms_all = D.sum(float(counts[1]))
    if D.rank != 0:
        return None
    tag, r = max(prof.items(), key=lambda kv: kv[1]["ms"])
    per_launch_s = r["ms"] / r["launches"] / 1e3
    achieved = r["bytes"] / r["launches"] / per_launch_s / 1e9 if r["bytes"] > 0 else 0.0
    roofline = {"kernel": tag, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": None, "avg_launch_ms": round(r["ms"] / r["launches"], 4), "launches": r["launches"],
                "kernels_ms_per_step": {k: round(v["ms"] / steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
                "hbm_gbs_by_kernel": {k: round(v["bytes"] / (v["ms"] / 1e3) / 1e9, 1) for k, v in prof.items() if v["bytes"] > 0 and v["ms"] > 0}}
    return {
        "metric": "3840x2160 frames/sec pyramid+gradient+NMS+SURF chain", "value": round(D.world * B * steps / elapsed, 2), "unit": "frames/s", "n_gpus": D.world,
        "steps": steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 pyramid/gradient/detect, f64 describe", "data": "synthetic",
        "config": {"workload": "batch of %d %dx%d GrayF32 per GPU, device resident: pyramid [1,2,4,8] (Gaussian r=2) -> Sobel -> |grad|^2 -> strict NMS r=2 on every "
                               "layer, Fast-Hessian + SURF-64 (stable) on layer 0" % (B, W, H), "batch_per_gpu": B, "width": W, "height": H,
                   "keypoints_per_frame": round(kp_all / (D.world * B), 1), "nms_maxima_per_frame": round(nms_all / (D.world * B), 1), "ranks_seen": D.ranks_seen()},
        "roofline": roofline, "cpu_baseline": None,
    }


# ---------------------------------------------------------------------------------------------------------------- workload: brief_frames
def run_brief_frames(args, D):
    """Fast-Hessian + BRIEF-512 (DetectDescribeFusion(fastHessian, null, brief)) on a device-resident batch, then greedy Hamming
    association of every frame with the next one on the words still resident: the BRIEF form of the default workload."""
    import ctypes as C
    import numpy as np
    torch = D.torch
    from boofcv_amd import api, _lib
    L = _lib.load()
    B = args.batch or 256
    H = args.height or 1080
    W = args.width or 1920
    frames = synth_frames(B, H, W, 1000 + D.rank * B, D.device)
    ctx = api.Context(D.local_rank, stream=torch.cuda.current_stream(D.local_rank).cuda_stream)
    brief = api.FactoryDescribeRegionPoint.brief(None, api.GrayF32, ctx=ctx)
    dd = api.FactoryDetectDescribe.fuseTogether(api.FactoryInterestPoint.fastHessian(None), None, brief, ctx=ctx)
    out = {"pairs": None, "fit": None, "total": 0}
    LL, I = C.POINTER(C.c_longlong), C.POINTER(C.c_int)

    def step():
        dd.detectDevice(frames.data_ptr(), H * W, W, W, H, B)
        total = dd.totalFeatures()
        if out["pairs"] is None or out["pairs"].numel() < total:
            out["pairs"] = torch.empty(max(total, 1) * 2, dtype=torch.int32, device=frames.device)
            out["fit"] = torch.empty(max(total, 1) * 2, dtype=torch.float64, device=frames.device)
        counts = dd.counts()
        starts = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        base = dd.deviceViewBrief(0)[0]
        dst_idx = (np.arange(B) + 1) % B
        so, do_ = np.ascontiguousarray(starts[:B]), np.ascontiguousarray(starts[dst_idx])
        ns, nd = np.ascontiguousarray(counts), np.ascontiguousarray(counts[dst_idx])
        st = L.bhip_assoc_hamming_dev_batched(ctx._h, C.c_void_p(base), C.c_void_p(base), 16, B, so.ctypes.data_as(LL), ns.ctypes.data_as(I), do_.ctypes.data_as(LL),
                                              nd.ctypes.data_as(I), api.Double_MAX_VALUE, 1, C.c_void_p(out["pairs"].data_ptr()), C.c_void_p(out["fit"].data_ptr()))
        if st != 0:
            raise RuntimeError("bhip_assoc_hamming_dev_batched failed: %s" % L.bhip_last_error(ctx._h))
        out["total"] = total
        return total

    for _ in range(max(args.warmup, 0)):
        step()
    warm = args.warmup
    args.warmup = 0
    ctx.profile(True); ctx.profileReset()
    if args.steps is None:
        D.barrier()
        t0 = time.perf_counter(); step(); torch.cuda.synchronize()
        one = D.max(time.perf_counter() - t0)
        args.steps = int(D.max(float(max(3, int(MIN_TIMED_SECONDS / max(one, 1e-6)) + 1))))
        ctx.profileReset()
    steps, elapsed, _ = timed_steps(D, args, step)
    args.warmup = warm
    prof = ctx.profileReport(); ctx.profile(False)
    kp_all = D.sum(float(out["total"]))
    matched = D.sum(float((out["pairs"][:out["total"]] >= 0).sum().item()))
    if D.rank != 0:
        return None
    tag, r = max(prof.items(), key=lambda kv: kv[1]["ms"])
    per_launch_s = r["ms"] / r["launches"] / 1e3
    if r["bytes"] > 0:
        achieved = r["bytes"] / r["launches"] / per_launch_s / 1e9
        roofline = {"kernel": tag, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None}
    else:
        tops = r["flops"] / r["launches"] / per_launch_s / 1e12 if r["flops"] > 0 else 0.0
        roofline = {"kernel": tag, "bound": "valu", "achieved": round(tops, 2), "peak": None, "unit": "T popcount-ops/s", "frac": None, "traffic": None}
    roofline.update({"avg_launch_ms": round(r["ms"] / r["launches"], 4), "launches": r["launches"],
                     "kernels_ms_per_step": {k: round(v["ms"] / steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}})
    return {
        "metric": "1080p frames/sec Fast-Hessian detect + BRIEF-512 describe + Hamming associate", "value": round(D.world * B * steps / elapsed, 2), "unit": "frames/s",
        "n_gpus": D.world, "steps": steps, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32 detect / u32 words, popcount Hamming", "data": "synthetic",
        "config": {"workload": "batch of %d %dx%d GrayF32 per GPU, device resident: Fast-Hessian + BRIEF-512 (fuseTogether(fastHessian, null, brief)) + greedy Hamming "
                               "associate with the next frame on the resident words" % (B, W, H), "batch_per_gpu": B, "width": W, "height": H,
                   "keypoints_per_frame": round(kp_all / (D.world * B), 1), "matches_per_frame": round(matched / (D.world * B), 1), "ranks_seen": D.ranks_seen()},
        "roofline": roofline, "cpu_baseline": None,
    }


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus < 1:
        sys.exit("--gpus must be >= 1")
    under_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    if not under_launcher and args.gpus > 1:
        sys.exit(launch_ranks(args, argv))      # no GPU call has been made in this process
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report a number for the wrong GPU count\n" % (args.gpus, world))
        sys.exit(2)
    if args.dry_launch:
        line = {"dry_launch": True, "rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "world_size": world, "n_gpus": args.gpus,
                "master": "%s:%s" % (os.environ.get("MASTER_ADDR", ""), os.environ.get("MASTER_PORT", "")), "pid": os.getpid()}
        # rank 0's line goes to stdout (relayed by the parent); the others identify themselves on stderr
        (sys.stdout if rank == 0 else sys.stderr).write(json.dumps(line) + "\n")
        return
    D = Dist(args)
    line = {"frames": run_frames, "assoc_sharded": run_assoc_sharded, "chain4k": run_chain4k, "brief_frames": run_brief_frames}[args.workload](args, D)
    if D.rank == 0:
        assert line["n_gpus"] == args.gpus
        if D.rehearsal:
            line["rehearsal"] = "ranks share the visible GPU(s), control collectives over gloo: exercises the rank plumbing, NOT a multi-GPU measurement"
        print(json.dumps(line), flush=True)
    D.close()


if __name__ == "__main__":
    main()
