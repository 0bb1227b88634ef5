# instruction mix, LDS activity and bank conflicts of the tiled run-time-width convolution kernels (41 taps, 64 x 1080p)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_convw && mkdir -p gpurun_out/pmc_convw
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_INSTS_SMEM" "SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS"; do
  tag=$(echo $set | cut -d' ' -f2)
  timeout -k 10 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmc_convw/$tag -- python3 scripts/probe/conv_wide_only.py > gpurun_out/pmc_convw/$tag.log 2>&1 || echo "pass $tag failed"
done
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob('gpurun_out/pmc_convw/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_conv_' not in k: continue
        k = k.split('(')[0].replace('void ', '')
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']] += 1
for k, v in agg.items():
    w = max(v.get('SQ_WAVES', 1), 1) / 2.0
    print(k, {c: round(x / (w if c != 'SQ_WAVES' else 1), 1) for c, x in sorted(v.items())}, '(per wave)')
PY
find gpurun_out/pmc_convw -name "*.csv" -size +2M -delete
