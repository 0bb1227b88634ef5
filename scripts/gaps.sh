# where is the GPU idle inside a step?  rocprofv3 kernel trace of 4 steps; gaps between consecutive kernels of the last steps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export BHIP_BENCH_FRAMES_CACHE=/tmp/bhip_frames
python3 bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-end-to-end --no-conv > /dev/null 2>&1
rm -rf gpurun_out/gaps && mkdir -p gpurun_out/gaps
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaps -- python3 bench.py --steps 4 --warmup 2 --cpu-frames 0 --no-end-to-end --no-conv > gpurun_out/gaps/bench.json 2> gpurun_out/gaps/stderr.log
python3 - <<'PY'
import csv, glob
rows = []
for f in glob.glob('gpurun_out/gaps/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '')[:40]))
rows.sort()
# last integral launch = start of the last step
starts = [i for i, r in enumerate(rows) if 'k_integral_fused' in r[2]]
i0 = starts[-2]; i1 = starts[-1]
step = rows[i0:i1 + 1]
print('step span %.3f ms, kernels %d, busy %.3f ms' % ((step[-1][0] - step[0][0]) / 1e6, len(step) - 1, sum(e - s for s, e, _ in step[:-1]) / 1e6))
for (s, e, n), (s2, e2, n2) in zip(step[:-1], step[1:]):
    gap = (s2 - e) / 1e3
    if gap > 8:
        print('  gap %7.1f us after %-40s before %s' % (gap, n, n2))
PY
find gpurun_out/gaps -name "*.csv" -size +1M -delete
