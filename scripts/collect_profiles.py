#!/usr/bin/env python3
"""gpurun_out/ (scripts/profile_all.sh and the per-phase PMC scripts) -> profiles/rNN_*: the committed evidence.  usage: collect_profiles.py NN"""
import csv
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1]
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")


def copy(src, name):
    src = os.path.join(G, src)
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copyfile(src, os.path.join(P, "r%s_%s" % (rnd, name)))
        print("r%s_%s" % (rnd, name))
    else:
        print("MISSING", src)


def json_line(src, name):
    """the one JSON line a bench run printed (anything else on stdout is dropped), pretty enough to diff"""
    path = os.path.join(G, src)
    if not os.path.exists(path):
        print("MISSING", path); return
    for line in open(path):
        line = line.strip()
        if line.startswith("{"):
            json.dump(json.loads(line), open(os.path.join(P, "r%s_%s" % (rnd, name)), "w"), indent=1)
            print("r%s_%s" % (rnd, name)); return
    print("NO JSON LINE in", path)


json_line("bench_default.json", "bench_default_run.json")
json_line("bench_brief_frames.json", "bench_brief_frames.json")
json_line("bench_chain4k.json", "bench_chain4k.json")
json_line("bench_assoc_sharded.json", "bench_assoc_sharded.json")
copy("conv_roofline.json", "conv_roofline.json")
copy("pmc_insts/summary.txt", "pmc_insts.txt")
copy("pmc_ta.log", "pmc_ta.txt")
copy("pmc_dphase/summary.txt", "pmc_describe_phases.txt")
copy("pmc_fphase/summary.txt", "pmc_fused_phases.txt")
copy("r3_gather_rate.txt", "gather_rate_probe.txt")
# rocprofv3 --kernel-trace --stats of the bench command: own kernels only (torch's input-synthesis kernels are not the product)
ks = os.path.join(G, "prof", "kernel_stats.csv")
if os.path.exists(ks):
    rows = list(csv.reader(open(ks)))
    own = [rows[0]] + [r for r in rows[1:] if r and (r[0].startswith(("k_", "void k_")) or "pyramid" in r[0])]
    csv.writer(open(os.path.join(P, "r%s_bench_kernel_stats.csv" % rnd), "w")).writerows(own)
    print("r%s_bench_kernel_stats.csv" % rnd, len(own) - 1, "kernels")
else:
    print("MISSING", ks)
if os.path.exists(os.path.join(G, "pmc_traffic", "summary.json")):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "make_pmc_profile.py"), rnd])
