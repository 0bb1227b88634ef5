# everything profiles/rNN_* is made from, in one GPU call (results under gpurun_out/, copied to profiles/ by scripts/collect_profiles.py)
set -x
timeout -k 10 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
timeout -k 10 600 bash scripts/profile_bench.sh > gpurun_out/profile_bench.log 2>&1
timeout -k 10 600 bash scripts/pmc_traffic.sh > gpurun_out/pmc_traffic.log 2>&1
timeout -k 10 600 bash scripts/pmc_insts.sh > gpurun_out/pmc_insts.log 2>&1
timeout -k 10 600 bash scripts/pmc_ta.sh > gpurun_out/pmc_ta.log 2>&1
timeout -k 10 300 python scripts/bench_conv.py --json gpurun_out/conv_roofline.json > gpurun_out/bench_conv.log 2>&1
timeout -k 10 300 python bench.py --workload brief_frames > gpurun_out/bench_brief_frames.json 2>/dev/null
timeout -k 10 300 python bench.py --workload chain4k > gpurun_out/bench_chain4k.json 2>/dev/null
timeout -k 10 300 python bench.py --workload assoc_sharded > gpurun_out/bench_assoc_sharded.json 2>/dev/null
ls -la gpurun_out | tail -20
