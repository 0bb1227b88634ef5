import sys, numpy as np
sys.path.insert(0, '.')
from boofcv_amd import api
w, h, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dd = api.FactoryDetectDescribe.surfFast(api.ConfigFastHessian(detectThreshold=1e12), None, None, api.GrayF32)
rng = np.random.default_rng(1)
f = rng.uniform(0, 100, (h, w)).astype(np.float32)
dd.detectBatch([api.GrayF32.wrap(f)] * n)
ii = dd.fetchIntegral(n - 1, w, h)
print("integral ok", float(ii[-1, -1]), flush=True)
