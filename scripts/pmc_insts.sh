# dynamic instruction mix per own kernel (rocprofv3 PMC, SQ counters), batch 32
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# synthetic frames are generated once outside the profiler (bench.py BHIP_BENCH_FRAMES_CACHE) and re-read by the profiled runs
export BHIP_BENCH_FRAMES_CACHE=/tmp/bhip_frames
python3 bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-end-to-end --no-conv --batch 32 > /dev/null 2>&1
rm -rf gpurun_out/pmc_insts && mkdir -p gpurun_out/pmc_insts
echo "pmc insts: $(date +%T)"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc_insts -- python3 bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-end-to-end --no-conv --batch 32 > gpurun_out/pmc_insts/run.log 2>&1
python3 - > gpurun_out/pmc_insts/summary.txt <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('gpurun_out/pmc_insts/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if not (k.startswith('k_') or k.startswith('void k_')):
            continue
        agg[k.split('(')[0].replace('void ', '')[:48]][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', 0)):
    w = max(v.get('SQ_WAVES', 1), 1)
    print('%-50s waves %9d  per wave: VALU %7.0f SALU %6.0f LDS %6.0f VMEMrd %5.0f VMEMwr %4.0f SMEM %4.0f  wave-cycles(x4) %8.0f' % (
        k, w, v.get('SQ_INSTS_VALU', 0) / w, v.get('SQ_INSTS_SALU', 0) / w, v.get('SQ_INSTS_LDS', 0) / w, v.get('SQ_INSTS_VMEM_RD', 0) / w,
        v.get('SQ_INSTS_VMEM_WR', 0) / w, v.get('SQ_INSTS_SMEM', 0) / w, 4 * v.get('SQ_WAVE_CYCLES', 0) / w))
PY
find gpurun_out/pmc_insts -name "*.csv" -size +2M -delete
cat gpurun_out/pmc_insts/summary.txt
# matrix-core counters of the association kernels (their own pass: SQ counters per pass are limited)
echo "pmc mfma: $(date +%T)"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INSTS_MFMA SQ_WAVES --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -- python3 bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-end-to-end --no-conv --batch 32 > gpurun_out/pmc_insts/run_mfma.log 2>&1
python3 - >> gpurun_out/pmc_insts/summary.txt <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob('gpurun_out/pmc_mfma/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'mfma' not in k:
            continue
        agg[k.split('(')[0].replace('void ', '')[:48]][r['Counter_Name']] += float(r['Counter_Value'])
print('matrix-core counters (sums over the launch):')
for k, v in sorted(agg.items()):
    busy = v.get('SQ_BUSY_CYCLES', 0)
    print('%-30s MFMA insts %10.0f  MOPS F16 %12.0f  I8 %12.0f  MFMA busy cycles %12.0f  SQ busy cycles %12.0f  ratio %.3f' % (
        k, v.get('SQ_INSTS_MFMA', 0), v.get('SQ_INSTS_VALU_MFMA_MOPS_F16', 0), v.get('SQ_INSTS_VALU_MFMA_MOPS_I8', 0), v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0), busy,
        v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / busy if busy else 0.0))
PY
find gpurun_out/pmc_mfma -name "*.csv" -size +2M -delete
tail -5 gpurun_out/pmc_insts/summary.txt
