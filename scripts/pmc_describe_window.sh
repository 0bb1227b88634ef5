# diagnostic: dynamic instruction counts inside k_describe's window sweep (experiments build): runs that return at the sweep's internal
# boundaries 8 (c(a) done), 9 (schedule validated), 10 (prefix sums), 11 (candidates), 12 (reduced), all stopping after the orientation (STOP=3)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export BHIP_LIB=$GRAFT_REPO_ROOT/boofcv_amd/libboofhip_exp.so
export BHIP_BENCH_FRAMES_CACHE=/tmp/bhip_frames
python3 bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-end-to-end --no-conv --batch 32 > /dev/null 2>&1
rm -rf gpurun_out/pmc_dwin && mkdir -p gpurun_out/pmc_dwin
export BHIP_DESCRIBE_STOP=3
for ws in 2 8 9 10 11 12 99; do
  if [ $ws = 2 ]; then export BHIP_DESCRIBE_STOP=2; export BHIP_DESCRIBE_WSTOP=-1; else export BHIP_DESCRIBE_STOP=3; export BHIP_DESCRIBE_WSTOP=$ws; fi
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_dwin/s$ws -- python3 bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-end-to-end --no-conv --batch 32 > gpurun_out/pmc_dwin/run$ws.log 2>&1
  python3 - $ws >> gpurun_out/pmc_dwin/summary.txt <<'PY'
import csv, glob, collections, sys
ws = sys.argv[1]
agg = collections.defaultdict(float)
for f in glob.glob('gpurun_out/pmc_dwin/s%s/**/*counter_collection.csv' % ws, recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_describe' in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value'])
w = max(agg.get('SQ_WAVES', 1), 1)
print('window stop %3s  per wave (cumulative): VALU %7.0f SALU %6.0f LDS %6.0f  wave-cycles(x4) %8.0f' % (
    ws, agg.get('SQ_INSTS_VALU', 0) / w, agg.get('SQ_INSTS_SALU', 0) / w, agg.get('SQ_INSTS_LDS', 0) / w, 4 * agg.get('SQ_WAVE_CYCLES', 0) / w))
PY
  find gpurun_out/pmc_dwin/s$ws -name "*.csv" -size +2M -delete
done
cat gpurun_out/pmc_dwin/summary.txt
