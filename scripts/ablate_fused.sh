# diagnostic: time of the fused detect kernels with phases removed (BHIP_FUSED_ABLATE bits: 1 intensity, 2 NMS, 4 staging); results are garbage
# needs the experiments build: python -m boofcv_amd.build --experiments (libboofhip_exp.so); the shipped library has none of these switches
export BHIP_LIB=${GRAFT_REPO_ROOT:-$PWD}/boofcv_amd/libboofhip_exp.so
for a in 0 4 1 2 3 5 6; do BHIP_FUSED_ABLATE=$a timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-end-to-end --no-conv --batch 64 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print('ablate $a', k.get('k_detect_fused_skip1'), k.get('k_detect_fused_skipN'))"; done
