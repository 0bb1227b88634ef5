// K4 + K5: region orientation and SURF descriptors, one wavefront per key point.
//
// Reference:
//   WrapDetectDescribeSurf.computeDescriptors            F:abst/feature/detdesc/WrapDetectDescribeSurf.java:116-128
//   OrientationIntegralBase                              F:alg/feature/orientation/OrientationIntegralBase.java:75-103
//   ImplOrientationSlidingWindowIntegral.compute/estimateAngle   F:alg/feature/orientation/impl/ImplOrientationSlidingWindowIntegral.java:81-188
//   ImplOrientationAverageGradientIntegral.compute       F:alg/feature/orientation/impl/ImplOrientationAverageGradientIntegral.java:54-127
//   SparseIntegralGradient_NoBorder(_F32)                I:alg/transform/ii/SparseIntegralGradient_NoBorder.java:42-47, impl/..._F32.java:46-76
//   SparseScaleGradient.isInBounds                       T:struct/sparse/SparseScaleGradient.java:48-50 ; SparseGradientSafe :56-61
//   DescribePointSurf.describe/features/computeLaplaceSign   F:alg/feature/describe/DescribePointSurf.java:169-313
//   DescribePointSurfMod.features                        F:alg/feature/describe/DescribePointSurfMod.java:121-192
//   UtilFeature.normalizeL2                              F:alg/descriptor/UtilFeature.java:101-114
//
// Arithmetic: gradients are fp32 box differences of the integral image (exact order kept), everything after that is fp64 in the
// reference's order with no FMA contraction.  The reference picks an unguarded sampler when SurfDescribeOps.isInside says the whole
// region is inside the image; the guarded sampler returns the same values there, so the device always samples guarded (never reads
// outside the image) and the isInside test disappears.  The 289 orientation angles are ordered by (angle, sample index): the
// reference's ddogleg QuickSort_F64 is unstable and its tie order is not pinned by any reference test (SURVEY hard part 5).
//
// Work split inside a wave: samples are spread over the 64 lanes and staged in LDS; the order-dependent fp64 reductions
// (sliding-window sweep, 81-term sub-region sums, L2 norm) each run on one lane per independent chain so they round exactly as
// the sequential Java loops do.
#include "common.h"

struct DescParams {
	ImgView ii;
	const KeyPoint* kps;      // [image][cap] or, when imageStart == nullptr, a flat list for image `singleImage`
	int cap;
	const int* imageStart;    // batch+1 exclusive prefix of per-image counts (device)
	int batch;
	int singleImage;
	long long total;
	SurfTables t;
	const double* anglesIn;   // optional: skip orientation and use these
	double* angles;           // [total]
	double* desc;             // [total][dof]
	uint8_t* white;           // [total]
	int ldsPerWave;           // bytes
	int gridW;                // descriptor sample grid width
};

// ---- sparse gradient (SparseIntegralGradient_NoBorder_F32) ----
__device__ __forceinline__ int gradRadius(double width) {
	int r = ((int)(width + 0.5)) / 2;
	if (r <= 0) r = 1;
	return r;
}
__device__ __forceinline__ bool gradInBounds(int x, int y, int r, int W, int H) { return x - r - 1 >= 0 && y - r - 1 >= 0 && x + r < W && y + r < H; }
__device__ __forceinline__ void gradCompute(const float* __restrict__ d, int stride, int x, int y, int r, float& gx, float& gy) {
	const int w = 2 * r + 1;
	const long long s1 = (long long)(y - r - 1) * stride + (x - r - 1);
	const long long s2 = s1 + (long long)r * stride;
	const long long s3 = s2 + stride;
	const long long s4 = s3 + (long long)r * stride;
	const float p0 = d[s1], p1 = d[s1 + r], p2 = d[s1 + r + 1], p3 = d[s1 + w];
	const float p11 = d[s2], p4 = d[s2 + w];
	const float p10 = d[s3], p5 = d[s3 + w];
	const float p9 = d[s4], p8 = d[s4 + r], p7 = d[s4 + r + 1], p6 = d[s4 + w];
	const float left = p8 - p9 - p1 + p0;
	const float right = p6 - p7 - p3 + p2;
	const float top = p4 - p11 - p3 + p0;
	const float bottom = p6 - p9 - p5 + p10;
	gx = right - left;
	gy = bottom - top;
}
__device__ __forceinline__ void gradSafe(const float* __restrict__ d, int stride, int W, int H, int x, int y, int r, float& gx, float& gy) {
	if (gradInBounds(x, y, r, W, H)) gradCompute(d, stride, x, y, r, gx, gy);
	else { gx = 0.0f; gy = 0.0f; }
}

// ---- clamped box sum (ImplIntegralImageOps.block_zero) for the Laplacian sign ----
__device__ __forceinline__ float blockZero(const float* __restrict__ d, int stride, int W, int H, int x0, int y0, int x1, int y1) {
	x0 = min(x0, W - 1); y0 = min(y0, H - 1); x1 = min(x1, W - 1); y1 = min(y1, H - 1);
	float br = 0, tr = 0, bl = 0, tl = 0;
	if (x1 >= 0 && y1 >= 0) br = d[(long long)y1 * stride + x1];
	if (y0 >= 0 && x1 >= 0) tr = d[(long long)y0 * stride + x1];
	if (x0 >= 0 && y1 >= 0) bl = d[(long long)y1 * stride + x0];
	if (x0 >= 0 && y0 >= 0) tl = d[(long long)y0 * stride + x0];
	return br - tr - bl + tl;
}

// georegression UtilAngle.dist: circular distance in [0,pi]
__device__ __forceinline__ double angleDist(double a, double b) {
	double diff = a - b;
	if (diff > M_PI) diff = diff - 2.0 * M_PI;
	else if (diff < -M_PI) diff = 2.0 * M_PI + diff;
	return fabs(diff);
}

__device__ __forceinline__ void waveSync() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(256) void k_describe(DescParams P) {
	extern __shared__ __attribute__((aligned(16))) unsigned char ldsAll[];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const long long g = (long long)blockIdx.x * 4 + wave;
	if (g >= P.total) return;
	unsigned char* lds = ldsAll + (size_t)wave * P.ldsPerWave;

	// which image does key point g belong to?
	int img, local;
	if (P.imageStart) {
		int lo = 0, hi = P.batch;  // find img with imageStart[img] <= g < imageStart[img+1]
		while (hi - lo > 1) {
			const int m = (lo + hi) >> 1;
			if ((long long)P.imageStart[m] <= g) lo = m; else hi = m;
		}
		img = lo;
		local = (int)(g - P.imageStart[img]);
	} else {
		img = P.singleImage;
		local = (int)g;
	}
	const KeyPoint kp = P.imageStart ? P.kps[(long long)img * P.cap + local] : P.kps[local];
	const float* __restrict__ d = P.ii.data + (long long)img * P.ii.imageStride;
	const int stride = P.ii.stride, W = P.ii.width, H = P.ii.height;
	const SurfTables& T = P.t;

	// ------------------------------------------------------------------ orientation
	double angle;
	if (P.anglesIn) {
		angle = P.anglesIn[g];
	} else {
		const double radius = kp.scale * 2.0;                 // BoofDefaults.SURF_SCALE_TO_RADIUS
		const double oscale = radius * T.oriRadiusToScale;    // setObjectRadius
		const int r = gradRadius(oscale * T.oriKernelWidth);
		const double period = oscale * T.oriPeriod;
		double tl_x = kp.x - T.oriRadius * period;
		double tl_y = kp.y - T.oriRadius * period;
		tl_x += 0.5;
		tl_y += 0.5;
		const int sw = T.oriWidth, n = sw * sw;
		double* dX = (double*)lds;
		double* dY = dX + n;
		double* ang = dY + n;
		int* order = (int*)(ang + n);
		for (int idx = lane; idx < n; idx += 64) {
			const int sy = idx / sw, sx = idx - sy * sw;
			const int xx = (int)(tl_x + sx * period);
			const int yy = (int)(tl_y + sy * period);
			float gx, gy;
			gradSafe(d, stride, W, H, xx, yy, r, gx, gy);
			double dx = (double)gx, dy = (double)gy;
			if (T.oriStable) {
				if (T.oriHasWeights) {
					const double w = T.oriWeights[idx];
					dx *= w;
					dy *= w;
				}
				dX[idx] = dx;
				dY[idx] = dy;
				ang[idx] = atan2(dy, dx);
			} else {
				// average variant accumulates w*gx (or gx) in row-major order; stage the addends
				if (T.oriHasWeights) {
					const double w = T.oriWeights[idx];
					dX[idx] = w * dx;
					dY[idx] = w * dy;
				} else {
					dX[idx] = dx;
					dY[idx] = dy;
				}
			}
		}
		waveSync();
		if (T.oriStable) {
			// arg-sort ascending by (angle, index): rank = number of elements ordered before mine
			for (int idx = lane; idx < n; idx += 64) {
				const double a = ang[idx];
				int rank = 0;
				for (int j = 0; j < n; j++) {
					const double b = ang[j];
					rank += (b < a || (b == a && j < idx)) ? 1 : 0;
				}
				order[rank] = idx;
			}
			waveSync();
			double bestX = 0, bestY = 0;
			if (lane == 0) {
				// estimateAngle(): sequential two-pointer sweep, exactly as written in the reference
				const int total = n;
				int start = 0, end = 1;
				int startIndex = order[start];
				int endIndex = order[end];
				double sumX = dX[startIndex], sumY = dY[startIndex];
				double best = sumX * sumX + sumY * sumY;
				bestX = sumX;
				bestY = sumY;
				double endAngle = ang[endIndex];
				const double window = T.oriWindow;
				while (start != total) {
					startIndex = order[start];
					const double startAngle = ang[startIndex];
					while (angleDist(startAngle, endAngle) <= window) {
						sumX += dX[endIndex];
						sumY += dY[endIndex];
						const double mag = sumX * sumX + sumY * sumY;
						if (mag > best) { best = mag; bestX = sumX; bestY = sumY; }
						end++;
						if (end >= total) end = 0;
						endIndex = order[end];
						endAngle = ang[endIndex];
						if (endIndex == startIndex) break;
					}
					sumX -= dX[startIndex];
					sumY -= dY[startIndex];
					start++;
				}
			}
			bestX = __shfl(bestX, 0, 64);
			bestY = __shfl(bestY, 0, 64);
			angle = atan2(bestY, bestX);
		} else {
			double Dx = 0, Dy = 0;
			if (lane == 0) {
				for (int i = 0; i < n; i++) { Dx += dX[i]; Dy += dY[i]; }
			}
			Dx = __shfl(Dx, 0, 64);
			Dy = __shfl(Dy, 0, 64);
			angle = atan2(Dy, Dx);
		}
		waveSync();
	}
	if (lane == 0 && P.angles) P.angles[g] = angle;
	if (!P.desc) return;

	// ------------------------------------------------------------------ descriptor
	const double c = cos(angle), s = sin(angle);
	const double scale = kp.scale;
	const int r = gradRadius(T.widthSample * scale);
	const int regionSize = T.widthLargeGrid * T.widthSubRegion;
	const int regionR = regionSize / 2;
	const int overLap = T.stable ? T.overLap : 0;
	const int gridW = regionSize + 2 * overLap;
	const int nsamp = gridW * gridW;
	float* sX = (float*)lds;
	float* sY = sX + nsamp;
	double* feat = (double*)(sY + nsamp + (nsamp & 1));  // keep 8-byte alignment
	{
		const double c_x = kp.x + 0.5, c_y = kp.y + 0.5;
		for (int idx = lane; idx < nsamp; idx += 64) {
			const int iy = idx / gridW, ix = idx - iy * gridW;
			const int rY = iy - regionR - overLap, rX = ix - regionR - overLap;
			const double regionY = rY * scale;
			const double regionX = rX * scale;
			const int pixelX = (int)(c_x + c * regionX - s * regionY);
			const int pixelY = (int)(c_y + s * regionX + c * regionY);
			float gx, gy;
			gradSafe(d, stride, W, H, pixelX, pixelY, r, gx, gy);
			sX[idx] = gx;
			sY[idx] = gy;
		}
	}
	waveSync();
	const int dof = T.dof;
	const int T_w = T.widthSubRegion + 2 * overLap;  // samples per sub-region side
	for (int f = lane; f < dof; f += 64) {
		const int sub = f >> 2, comp = f & 3;
		const int suby = sub / T.widthLargeGrid, subx = sub - suby * T.widthLargeGrid;
		const int rY = -regionR + suby * T.widthSubRegion, rX = -regionR + subx * T.widthSubRegion;
		double sum = 0;
		for (int i = 0; i < T_w; i++) {
			int index = (rY + regionR + i) * gridW + rX + regionR;
			for (int j = 0; j < T_w; j++, index++) {
				const double w = T.stable ? T.weightSub[i * T_w + j] : T.weightFast[(regionR + rY + i) * regionSize + regionR + rX + j];
				const double dx = w * (double)sX[index];
				const double dy = w * (double)sY[index];
				const double pdx = c * dx + s * dy;
				const double pdy = -s * dx + c * dy;
				const double v = comp < 2 ? pdx : pdy;
				sum += (comp & 1) ? fabs(v) : v;
			}
		}
		if (T.stable) sum = T.weightGrid[sub] * sum;
		feat[f] = sum;
	}
	waveSync();
	// normalizeL2: sequential sum of squares
	double norm = 0;
	if (lane == 0) {
		for (int i = 0; i < dof; i++) { const double v = feat[i]; norm += v * v; }
	}
	norm = __shfl(norm, 0, 64);
	double* out = P.desc + g * dof;
	if (norm == 0) {
		for (int f = lane; f < dof; f += 64) out[f] = feat[f];
	} else {
		norm = sqrt(norm);
		for (int f = lane; f < dof; f += 64) out[f] = feat[f] / norm;
	}
	// Laplacian sign (computeLaplaceSign): kernelDerivXX(9s) + kernelDerivYY(9s) at the rounded location
	if (lane == 0 && P.white) {
		const int x = (int)(kp.x + 0.5), y = (int)(kp.y + 0.5);
		const int si = (int)ceil(scale);
		const int size = 9 * si;
		const int blockW = size / 3, blockH = size - blockW - 1;
		const int r1 = blockW / 2, r2 = blockW + r1, r3 = blockH / 2;
		float xx = 0;
		xx += blockZero(d, stride, W, H, x - r2 - 1, y - r3 - 1, x + r2, y + r3) * 1.0f;
		xx += blockZero(d, stride, W, H, x - r1 - 1, y - r3 - 1, x + r1, y + r3) * -3.0f;
		float yy = 0;
		yy += blockZero(d, stride, W, H, x - r3 - 1, y - r2 - 1, x + r3, y + r2) * 1.0f;
		yy += blockZero(d, stride, W, H, x - r3 - 1, y - r1 - 1, x + r3, y + r1) * -3.0f;
		double lap = (double)xx;
		lap += (double)yy;
		P.white[g] = lap > 0 ? 1 : 0;
	}
}

int bhip_describe_lds_bytes(const SurfTables& t) {
	const int n = t.oriWidth * t.oriWidth;
	const int ori = n * (3 * 8 + 4);
	const int overLap = t.stable ? t.overLap : 0;
	const int gridW = t.widthLargeGrid * t.widthSubRegion + 2 * overLap;
	const int ns = gridW * gridW;
	const int desc = (2 * ns + (ns & 1)) * 4 + t.dof * 8;
	int b = ori > desc ? ori : desc;
	return (b + 15) & ~15;
}

int bhip_launch_describe_ex(bhip_ctx* ctx, ImgView ii, const KeyPoint* kps, int cap, const int* imageStart, int batch, int singleImage, long long total,
							SurfTables t, const double* anglesIn, double* angles, double* desc, uint8_t* white) {
	if (total <= 0) return BHIP_OK;
	DescParams P;
	P.ii = ii; P.kps = kps; P.cap = cap; P.imageStart = imageStart; P.batch = batch; P.singleImage = singleImage; P.total = total; P.t = t;
	P.anglesIn = anglesIn; P.angles = angles; P.desc = desc; P.white = white;
	P.ldsPerWave = bhip_describe_lds_bytes(t);
	P.gridW = 0;
	if (P.ldsPerWave * 4 > 160 * 1024) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "orientation/descriptor sample grid too large for LDS");
	const long long blocks = (total + 3) / 4;
	if (blocks > 0x7fffffffLL) return bhip_fail(ctx, BHIP_ERR_INVALID, "too many key points");
	{
		ProfScope ps(ctx, "k_describe");
		hipLaunchKernelGGL(k_describe, dim3((unsigned)blocks), dim3(256), (size_t)P.ldsPerWave * 4, ctx->stream, P);
	}
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}
