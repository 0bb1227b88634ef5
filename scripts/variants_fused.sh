# diagnostic: tile-shape variants of the fused detect kernels (BHIP_FUSED_VARIANT)
# needs the experiments build: python -m boofcv_amd.build --experiments (libboofhip_exp.so); the shipped library has none of these switches
export BHIP_LIB=${GRAFT_REPO_ROOT:-$PWD}/boofcv_amd/libboofhip_exp.so
for v in b a x y z w; do BHIP_FUSED_VARIANT=$v timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-end-to-end --batch 64 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print('variant $v', k.get('k_detect_fused_skip1'), k.get('k_detect_fused_skipN'), d['config']['keypoints_per_frame'])"; done
