# diagnostic: k_describe time against workgroups per CU (LDS padding lowers the occupancy)
# needs the experiments build: python -m boofcv_amd.build --experiments (libboofhip_exp.so); the shipped library has none of these switches
export BHIP_LIB=${GRAFT_REPO_ROOT:-$PWD}/boofcv_amd/libboofhip_exp.so
for pad in 0 10000 11800 12500 25000 38000 41000 60000 100000; do BHIP_DESCRIBE_LDSPAD=$pad timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-end-to-end --no-conv --batch 64 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pad $pad', d['roofline']['kernels_ms_per_step']['k_describe'])"; done
