# diagnostic: k_describe time against workgroups per CU (LDS padding lowers the occupancy; 40576 B per workgroup unpadded = 4 per CU)
# needs the experiments build: python -m boofcv_amd.build --experiments (libboofhip_exp.so); the shipped library has none of these switches
export BHIP_LIB=${GRAFT_REPO_ROOT:-$PWD}/boofcv_amd/libboofhip_exp.so
for pad in 0 8000 16000 45000 90000; do BHIP_DESCRIBE_LDSPAD=$pad timeout -k 10 200 python bench.py --steps 6 --warmup 2 --cpu-frames 0 --no-end-to-end --no-conv --batch 256 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pad $pad  k_describe ms', d['roofline']['kernels_ms_per_step']['k_describe'])"; done
