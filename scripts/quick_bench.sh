# short bench: kernel ms per step for the default workload at batch 64 (diagnostic)
timeout -k 10 300 python bench.py --steps ${STEPS:-8} --warmup 2 --cpu-frames 0 --no-end-to-end --no-conv --batch ${BATCH:-64} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('fps', d['value'], 'ms/step', d['ms_per_step'], 'kp/frame', d['config']['keypoints_per_frame'])
for k,v in d['roofline']['kernels_ms_per_step'].items(): print('  %-28s %8.3f' % (k, v))
"
