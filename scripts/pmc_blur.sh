# instruction mix and wave time of the one-pass Gaussian blur kernels (rocprofv3 PMC)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_blur && mkdir -p gpurun_out/pmc_blur
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f2)
  timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_blur/$tag -- python3 scripts/probe/blur_only.py > gpurun_out/pmc_blur/$tag.log 2>&1 || echo "pass $tag failed"
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('gpurun_out/pmc_blur/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_blur_fused' not in k: continue
        k = k.split('(')[0].replace('void ', '')
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
for k, v in agg.items():
    w = max(v.get('SQ_WAVES', 1), 1) / 2.0   # SQ_WAVES is collected in both passes
    print(k, 'launches', n[k]['SQ_INSTS_VALU'], {c: round(x / (w if c != 'SQ_WAVES' else 1), 1) for c, x in sorted(v.items())}, '(per wave)')
PY
find gpurun_out/pmc_blur -name "*.csv" -size +2M -delete
