#!/usr/bin/env python3
"""Experiment: does running two half-batches concurrently on two HIP streams (two contexts, two host threads) beat one stream over
the whole batch?  (describe is latency/VALU bound, the fused detector is LDS bound: co-scheduling could fill idle issue slots.)"""
import os, sys, threading, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

B = int(os.environ.get("BATCH", "128"))
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
frames = bench.synth_frames(B, 1080, 1920, 1000, dev)
torch.cuda.synchronize()


class HP(bench.HotPath):
    def __init__(self, batch, stream):
        from boofcv_amd import api, _lib
        self.api, self._lib = api, _lib
        self.L = _lib.load()
        self.ctx = api.Context(0, stream=stream.cuda_stream)
        self.dd = api.FactoryDetectDescribe.surfStable(None, None, None, api.GrayF32, ctx=self.ctx)
        self.batch, self.h, self.w = batch, 1080, 1920
        self.pairs = None; self.fit = None


def run_single(steps=4):
    s = torch.cuda.Stream()
    hp = HP(B, s)
    with torch.cuda.stream(s):
        hp.step(frames); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps): hp.step(frames)
        torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0)


def run_double(steps=4):
    half = B // 2
    ss = [torch.cuda.Stream(), torch.cuda.Stream()]
    hps = [HP(half, ss[0]), HP(half, ss[1])]
    parts = [frames[:half], frames[half:]]
    def work(i, n):
        with torch.cuda.stream(ss[i]):
            for _ in range(n): hps[i].step(parts[i])
            ss[i].synchronize()
    for i in range(2): work(i, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(i, steps)) for i in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0)


print("single stream  %.0f frames/s" % run_single())
print("two streams    %.0f frames/s" % run_double())
print("single stream  %.0f frames/s" % run_single())
