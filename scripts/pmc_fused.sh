cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc && mkdir -p gpurun_out/pmc
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc -- python3 bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-end-to-end --no-conv --batch 16 > gpurun_out/pmc/run.log 2>&1
python3 - <<'PY'
import csv, glob, collections
files = glob.glob('gpurun_out/pmc/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in files:
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:40]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in agg.items():
    if 'detect_fused' in k or 'hessian' in k or 'describe' in k:
        print(k, {a: int(b) for a, b in v.items()})
PY
