# rocprofv3 kernel-trace summary of the default bench command (copied to profiles/ by hand afterwards)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# synthetic frames are generated once outside the profiler (bench.py BHIP_BENCH_FRAMES_CACHE) and re-read by the profiled runs
export BHIP_BENCH_FRAMES_CACHE=/tmp/bhip_frames
python3 bench.py --steps 1 --warmup 0 --cpu-frames 0 --no-end-to-end --no-conv > /dev/null 2>&1
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-end-to-end --no-conv > gpurun_out/prof/bench_under_rocprof.json 2> gpurun_out/prof/stderr.log
find gpurun_out/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof/kernel_stats.csv
head -30 gpurun_out/prof/kernel_stats.csv
find gpurun_out/prof -name "*kernel_trace.csv" -delete
