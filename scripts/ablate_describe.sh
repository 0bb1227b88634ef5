# diagnostic: per-phase cycle shares of k_describe (BHIP_DESCRIBE_STAMPS) and its time with / without the serial window sweep
# needs the experiments build: python -m boofcv_amd.build --experiments (libboofhip_exp.so); the shipped library has none of these switches
export BHIP_LIB=${GRAFT_REPO_ROOT:-$PWD}/boofcv_amd/libboofhip_exp.so
rm -f gpurun_out/stamps.txt
BHIP_DESCRIBE_STAMPS=gpurun_out/stamps.txt timeout -k 10 200 python bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-end-to-end --no-conv --batch 64 > /dev/null 2>&1; cat gpurun_out/stamps.txt
for a in 0 1; do BHIP_DESCRIBE_SERIAL=$a timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-end-to-end --no-conv --batch 64 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('serial', $a, d['roofline']['kernels_ms_per_step']['k_describe'])"; done
