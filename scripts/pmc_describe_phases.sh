# diagnostic: dynamic instruction counts of k_describe per phase, by difference of runs that stop at successive phase boundaries
# (experiments build: python -m boofcv_amd.build --experiments).  Phases: 1 samples+atan2, 2 sort, 3 window, 4 descriptor samples, 5 sums, 6 rest
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export BHIP_LIB=$GRAFT_REPO_ROOT/boofcv_amd/libboofhip_exp.so
export BHIP_BENCH_FRAMES_CACHE=/tmp/bhip_frames
python3 bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-end-to-end --no-conv --batch 32 > /dev/null 2>&1
rm -rf gpurun_out/pmc_dphase && mkdir -p gpurun_out/pmc_dphase
for stop in 1 2 3 4 5 99; do
  export BHIP_DESCRIBE_STOP=$stop
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_dphase/s$stop -- python3 bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-end-to-end --no-conv --batch 32 > gpurun_out/pmc_dphase/run$stop.log 2>&1
  python3 - $stop >> gpurun_out/pmc_dphase/summary.txt <<'PY'
import csv, glob, collections, sys
stop = sys.argv[1]
agg = collections.defaultdict(float)
for f in glob.glob('gpurun_out/pmc_dphase/s%s/**/*counter_collection.csv' % stop, recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_describe' in r['Kernel_Name']:
            agg[r['Counter_Name']] += float(r['Counter_Value'])
w = max(agg.get('SQ_WAVES', 1), 1)
print('stop %3s  waves %8d  per wave: VALU %7.0f SALU %6.0f LDS %6.0f VMEMrd %5.0f SMEM %4.0f  wave-cycles(x4) %8.0f' % (
    stop, w, agg.get('SQ_INSTS_VALU', 0) / w, agg.get('SQ_INSTS_SALU', 0) / w, agg.get('SQ_INSTS_LDS', 0) / w, agg.get('SQ_INSTS_VMEM_RD', 0) / w,
    agg.get('SQ_INSTS_SMEM', 0) / w, 4 * agg.get('SQ_WAVE_CYCLES', 0) / w))
PY
  find gpurun_out/pmc_dphase/s$stop -name "*.csv" -size +2M -delete
done
cat gpurun_out/pmc_dphase/summary.txt
