# HBM-side traffic of every own kernel (rocprofv3 PMC, FETCH_SIZE and WRITE_SIZE in separate passes as the TCC slots require).
# Writes gpurun_out/pmc_traffic/summary.json: per kernel, counter sums / launches.  Batch is the bench default (256).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# synthetic frames are generated once outside the profiler (bench.py BHIP_BENCH_FRAMES_CACHE) and re-read by the profiled runs
export BHIP_BENCH_FRAMES_CACHE=/tmp/bhip_frames
python3 bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-end-to-end --no-conv > /dev/null 2>&1
rm -rf gpurun_out/pmc_traffic && mkdir -p gpurun_out/pmc_traffic
for c in FETCH_SIZE WRITE_SIZE; do
  echo "pmc pass $c: $(date +%T)"
  BHIP_BENCH_TRACE=1 timeout -k 10 180 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmc_traffic/$c -- python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-end-to-end --no-conv > gpurun_out/pmc_traffic/$c.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob('gpurun_out/pmc_traffic/%s/**/*counter_collection.csv' % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != c:
                continue
            k = r['Kernel_Name']
            if not (k.startswith('k_') or k.startswith('void k_')):
                continue
            k = k.split('(')[0].replace('void ', '')
            agg[k][0] += float(r['Counter_Value']); agg[k][1] += 1
    for k, (v, n) in agg.items():
        out.setdefault(k, {})[c] = {"sum": v, "launches": n, "per_launch": v / n}
json.dump(out, open('gpurun_out/pmc_traffic/summary.json', 'w'), indent=1)
for k, v in sorted(out.items()):
    print(k, {c: round(x["per_launch"], 1) for c, x in v.items()})
PY
find gpurun_out/pmc_traffic -name "*.csv" -size +2M -delete
