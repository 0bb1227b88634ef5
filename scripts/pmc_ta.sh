# texture-addresser / L1 pressure of the gather kernels (rocprofv3 PMC, batch 32): TA busy share, requests per wave, stalls
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export BHIP_BENCH_FRAMES_CACHE=/tmp/bhip_frames
python3 bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-end-to-end --no-conv --batch 32 > /dev/null 2>&1
rm -rf gpurun_out/pmc_ta && mkdir -p gpurun_out/pmc_ta
for set in "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_ta/$tag -- python3 bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-end-to-end --no-conv --batch 32 > gpurun_out/pmc_ta/$tag.log 2>&1 || echo "pass $tag failed"
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('gpurun_out/pmc_ta/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if not (k.startswith('k_') or k.startswith('void k_')):
            continue
        k = k.split('(')[0].replace('void ', '')[:44]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[k][r['Counter_Name']] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0)):
    if v.get('GRBM_GUI_ACTIVE', 0) < 1e5:
        continue
    print(k, {c: (round(x / cnt[k][c], 1) if 'avr' in c or 'max' in c else int(x)) for c, x in sorted(v.items())})
PY
find gpurun_out/pmc_ta -name "*.csv" -size +2M -delete
