# diagnostic: dynamic instruction counts of the fused detect kernels with phases removed (BHIP_FUSED_ABLATE bits: 1 intensity, 2 NMS, 4 staging);
# differences against the full run give the per-phase counts.  Experiments build only (python -m boofcv_amd.build --experiments).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export BHIP_LIB=$GRAFT_REPO_ROOT/boofcv_amd/libboofhip_exp.so
export BHIP_BENCH_FRAMES_CACHE=/tmp/bhip_frames
python3 bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-end-to-end --no-conv --batch 32 > /dev/null 2>&1
rm -rf gpurun_out/pmc_fphase && mkdir -p gpurun_out/pmc_fphase
for ab in 0 1 2 4 7; do
  export BHIP_FUSED_ABLATE=$ab
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmc_fphase/a$ab -- python3 bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-end-to-end --no-conv --batch 32 > gpurun_out/pmc_fphase/run$ab.log 2>&1
  python3 - $ab >> gpurun_out/pmc_fphase/summary.txt <<'PY'
import csv, glob, collections, sys
ab = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('gpurun_out/pmc_fphase/a%s/**/*counter_collection.csv' % ab, recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_detect_fused' in r['Kernel_Name']:
            key = 'skip1' if 'Li1ELi9E' in r['Kernel_Name'] or 'float, 1, 9' in r['Kernel_Name'] else 'skip2'
            agg[key][r['Counter_Name']] += float(r['Counter_Value'])
for key, v in sorted(agg.items()):
    w = max(v.get('SQ_WAVES', 1), 1)
    print('ablate %s %s waves %8d  per wave: VALU %7.0f SALU %6.0f LDS %6.0f VMEMrd %5.0f SMEM %4.0f  LDS bank-conflict cycles %7.0f  wave-cycles(x4) %8.0f' % (
        ab, key, w, v.get('SQ_INSTS_VALU', 0) / w, v.get('SQ_INSTS_SALU', 0) / w, v.get('SQ_INSTS_LDS', 0) / w, v.get('SQ_INSTS_VMEM_RD', 0) / w,
        v.get('SQ_INSTS_SMEM', 0) / w, v.get('SQ_LDS_BANK_CONFLICT', 0) / w, 4 * v.get('SQ_WAVE_CYCLES', 0) / w))
PY
  find gpurun_out/pmc_fphase/a$ab -name "*.csv" -size +2M -delete
done
cat gpurun_out/pmc_fphase/summary.txt
