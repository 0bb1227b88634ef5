#!/usr/bin/env python3
"""gpurun_out/pmc_traffic/summary.json (scripts/pmc_traffic.sh) -> profiles/rNN_pmc_traffic.json in the layout bench.py's pmc_traffic() reads.
usage: make_pmc_profile.py NN"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
s = json.load(open(os.path.join(ROOT, "gpurun_out", "pmc_traffic", "summary.json")))
kernels = {}
for k, v in sorted(s.items()):
    f = v.get("FETCH_SIZE", {}).get("per_launch", 0.0)
    w = v.get("WRITE_SIZE", {}).get("per_launch", 0.0)
    kernels[k] = {"fetch_kib": round(f, 1), "write_kib": round(w, 1), "bytes_per_launch": int((2 * f + w) * 1024)}


def find(prefix):
    return next((k for k in kernels if k.startswith(prefix)), None)


tags = {"k_describe": find("k_describe<false"), "k_detect_fused_skip1": find("k_detect_fused_fixed<float, 1,"),
        "k_detect_fused_skipN": find("k_detect_fused_fixed<float, 2,"), "k_hessian_skipN": find("k_hessian<"),
        "k_assoc_mfma_pass1": find("k_assoc_mfma<1>"), "k_assoc_mfma_pass2": find("k_assoc_mfma<2>"), "k_integral_fused": find("k_integral_fused")}
out = {
    "_about": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, scripts/pmc_traffic.sh) of `python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 "
              "--no-end-to-end` (256 x 1920x1080 per launch) on MI355X. Counters are KiB per launch. Corrections (MI355X_MICROARCH.md, HBM section): "
              "FETCH_SIZE reports half of the bytes on gfx950 -- checked against k_integral_fused, whose known 2.123 GB read shows as 1.05e6 KiB -- so "
              "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024; WRITE_SIZE is exact (integral: 2.0746e6 KiB for 2.123 GB). Infinity-Cache hits are counted, "
              "so this is traffic leaving the XCD L2s, an upper bound on HBM bytes.",
    "workload": {"batch": 256, "width": 1920, "height": 1080},
    "bench_tags": {k: v for k, v in tags.items() if v},
    "kernels": kernels,
}
path = os.path.join(ROOT, "profiles", "r%s_pmc_traffic.json" % rnd)
json.dump(out, open(path, "w"), indent=1)
print(path, len(kernels), "kernels")
