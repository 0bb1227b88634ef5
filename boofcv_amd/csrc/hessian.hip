// K2: Fast-Hessian determinant-of-Hessian intensity from the integral image, bit-exact with the reference.
//
// Reference: IntegralImageFeatureIntensity.hessian  F:alg/feature/detect/intensity/IntegralImageFeatureIntensity.java:43-56
//   -> hessianBorder  F:alg/feature/detect/intensity/impl/ImplIntegralImageFeatureIntensity.java:72-113  (clamped box sums)
//   -> hessianInner   ...:132-213  (32 integral-image taps per pixel)
//   box kernels       I:alg/transform/ii/DerivativeIntegralImage.java:102-158
//   block_zero        I:alg/transform/ii/impl/ImplIntegralImageOps.java:195-214, convolveSparse :172-183
// Every pixel of every level of one octave is one thread; all levels of the octave share one launch so the integral image
// is pulled through L2 once per octave.  fp32, no FMA contraction (-ffp-contract=off), expressions in the reference's order.
// Bound: HBM (+L2 gather).  Algorithmic bytes per octave: 4P (ii read) + levels * 4P/skip^2 (intensity write).
#include "common.h"

struct HessLevel {
	int size;
	int bS, bL, rF, rS;     // blockSmall, blockLarge, radiusFeature, radiusSkinny
	int border, lost;       // border (in output pixels), lostPixel
	float norm;
	// kernelDerivXX / YY / XY parameters for the border path
	int r1, r2, r3, b;
};

struct HessParams {
	ImgView ii;
	int skip, w, h, nlevels;
	float* out;             // [image][level][h][outStride]
	long long levelStride, imageStrideOut;
	int outStride;
	HessLevel lv[BHIP_MAX_LEVELS];
	HessLevelSource from[BHIP_MAX_LEVELS];
	int srcBorder[BHIP_MAX_LEVELS];   // shared levels: the producing octave's (step skip/2) inner-region border, width and height
	int srcW, srcH;
};

// T = float (GrayF32 integral image) or int (GrayS32: exact integer box sums, converted where the reference converts -- at the
// assignment / compound assignment into a float: ImplIntegralImageFeatureIntensity.java:245-390)
template <class T>
__device__ __forceinline__ T block_zero(const T* __restrict__ d, int stride, int W, int H, int x0, int y0, int x1, int y1) {
	x0 = min(x0, W - 1);
	y0 = min(y0, H - 1);
	x1 = min(x1, W - 1);
	y1 = min(y1, H - 1);
	// branch-free: the four corners are always fetched (from coordinates clamped into the image) and zeroed afterwards, so the
	// 40 taps of a border pixel are independent loads in flight together
	const int cx0 = max(x0, 0), cy0 = max(y0, 0), cx1 = max(x1, 0), cy1 = max(y1, 0);
	const T vbr = d[(long long)cy1 * stride + cx1], vtr = d[(long long)cy0 * stride + cx1];
	const T vbl = d[(long long)cy1 * stride + cx0], vtl = d[(long long)cy0 * stride + cx0];
	const T br = (x1 >= 0 && y1 >= 0) ? vbr : T(0);
	const T tr = (y0 >= 0 && x1 >= 0) ? vtr : T(0);
	const T bl = (x0 >= 0 && y1 >= 0) ? vbl : T(0);
	const T tl = (x0 >= 0 && y0 >= 0) ? vtl : T(0);
	return br - tr - bl + tl;
}

template <class T>
__global__ __launch_bounds__(256) void k_hessian(HessParams P) {
	const int x = blockIdx.x * blockDim.x + threadIdx.x;
	const int y = blockIdx.y;
	const int img = blockIdx.z / P.nlevels;
	const int level = blockIdx.z - img * P.nlevels;
	if (x >= P.w) return;
	const HessLevel L = P.lv[level];
	const bool inner = x >= L.border && x < P.w - L.border && y >= L.border && y < P.h - L.border;
	if (P.from[level].src) {
		// Same kernel size one octave down.  A response depends on the pixel and the kernel size, and on which of the reference's two
		// forms evaluates it: hessianInner sums the Dyy boxes as ((br - bl) - tr) + tl, the border form (block_zero) as ((br - tr) - bl) + tl,
		// so the two can differ in the last bit.  The inner regions of the two octaves are not the same set of pixels (borderOrig depends
		// on the step), hence: copy where both octaves use the same form, compute in place on the few rows / columns where they do not.
		const int bp = P.srcBorder[level];
		const int px = 2 * x, py = 2 * y;
		const bool srcInner = px >= bp && px < P.srcW - bp && py >= bp && py < P.srcH - bp;
		if (srcInner == inner) {
			const HessLevelSource S = P.from[level];
			P.out[(long long)img * P.imageStrideOut + (long long)level * P.levelStride + (long long)y * P.outStride + x] =
				S.src[(long long)img * S.imageStride + (long long)(y * S.step) * S.stride + x * S.step];
			return;
		}
	}
	const T* __restrict__ d = (const T*)P.ii.data + (long long)img * P.ii.imageStride;
	const int stride = P.ii.stride;
	const int skip = P.skip;
	const int xx = x * skip, yy = y * skip;
	float Dxx, Dyy, Dxy;
	if (inner) {
		// hessianInner: the first inner column sits at offset `lost`, then +skip per output pixel
		const int col = L.lost + (x - L.border) * skip;
		const long long top = (long long)(yy - L.rS - 1) * stride + col;
		const long long bot = top + (long long)L.bL * stride;
		const int bS = L.bS;
		Dxx = (float)(d[bot + 3 * bS] - d[top + 3 * bS] - d[bot] + d[top]);
		Dxx -= (float)(T(3) * (d[bot + 2 * bS] - d[top + 2 * bS] - d[bot + bS] + d[top + bS]));

		const long long l = (long long)(yy - L.rF - 1) * stride + (L.rF - L.rS) + col;
		const long long r = l + L.bL;
		const long long ro1 = (long long)bS * stride;
		Dyy = (float)(d[r + 3 * ro1] - d[l + 3 * ro1] - d[r] + d[l]);
		Dyy -= (float)(T(3) * (d[r + 2 * ro1] - d[l + 2 * ro1] - d[r + ro1] + d[l + ro1]));

		const long long y1 = (long long)(yy - bS - 1) * stride + (L.rF - bS) + col;
		const long long y2 = y1 + ro1;
		const long long y3 = y2 + stride;
		const long long y4 = y3 + ro1;
		const int x3 = bS + 1, x4 = x3 + bS;
		Dxy = (float)(d[y2 + bS] - d[y1 + bS] - d[y2] + d[y1]);
		Dxy -= (float)(d[y2 + x4] - d[y1 + x4] - d[y2 + x3] + d[y1 + x3]);
		Dxy += (float)(d[y4 + x4] - d[y3 + x4] - d[y4 + x3] + d[y3 + x3]);
		Dxy -= (float)(d[y4 + bS] - d[y3 + bS] - d[y4] + d[y3]);
	} else {
		// computeHessian via convolveSparse: ret = 0; ret += block_zero(...) * scale, block by block (float scales for GrayF32, int for GrayS32)
		const int W = P.ii.width, H = P.ii.height;
		T ret = 0;
		ret += block_zero<T>(d, stride, W, H, xx - L.r2 - 1, yy - L.r3 - 1, xx + L.r2, yy + L.r3) * T(1);
		ret += block_zero<T>(d, stride, W, H, xx - L.r1 - 1, yy - L.r3 - 1, xx + L.r1, yy + L.r3) * T(-3);
		Dxx = (float)ret;
		ret = 0;
		ret += block_zero<T>(d, stride, W, H, xx - L.r3 - 1, yy - L.r2 - 1, xx + L.r3, yy + L.r2) * T(1);
		ret += block_zero<T>(d, stride, W, H, xx - L.r3 - 1, yy - L.r1 - 1, xx + L.r3, yy + L.r1) * T(-3);
		Dyy = (float)ret;
		ret = 0;
		const int b = L.b;
		ret += block_zero<T>(d, stride, W, H, xx - b - 1, yy - b - 1, xx - 1, yy - 1) * T(1);
		ret += block_zero<T>(d, stride, W, H, xx, yy - b - 1, xx + b, yy - 1) * T(-1);
		ret += block_zero<T>(d, stride, W, H, xx, yy, xx + b, yy + b) * T(1);
		ret += block_zero<T>(d, stride, W, H, xx - b - 1, yy, xx - 1, yy + b) * T(-1);
		Dxy = (float)ret;
	}
	Dxx *= L.norm;
	Dxy *= L.norm;
	Dyy *= L.norm;
	const float det = Dxx * Dyy - 0.81f * Dxy * Dxy;
	P.out[(long long)img * P.imageStrideOut + (long long)level * P.levelStride + (long long)y * P.outStride + x] = det;
}

static HessLevel makeLevel(int size, int skip) {
	HessLevel L;
	L.size = size;
	L.bS = size / 3;
	L.bL = size - L.bS - 1;
	L.rF = size / 2;
	L.rS = L.bL / 2;
	const int borderOrig = L.rF + 1 + (skip - (L.rF + 1) % skip);
	L.border = borderOrig / skip;
	L.lost = borderOrig - L.rF - 1;
	L.norm = 1.0f / (float)(size * size);
	const int blockW = size / 3, blockH = size - blockW - 1;
	L.r1 = blockW / 2;
	L.r2 = blockW + L.r1;
	L.r3 = blockH / 2;
	L.b = size / 3;
	return L;
}

int bhip_launch_hessian(bhip_ctx* ctx, ImgView ii, int batch, int skip, int nlevels, const int* sizes, float* intensity, long long levelStride,
						long long imageStrideOut, int outStride, const HessLevelSource* from, bool intTaps) {
	if (nlevels > BHIP_MAX_LEVELS) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "too many scales per octave");
	HessParams P;
	P.ii = ii;
	P.skip = skip;
	P.w = ii.width / skip;
	P.h = ii.height / skip;
	P.nlevels = nlevels;
	P.out = intensity;
	P.levelStride = levelStride;
	P.imageStrideOut = imageStrideOut;
	P.outStride = outStride;
	for (int i = 0; i < nlevels; i++) {
		P.lv[i] = makeLevel(sizes[i], skip);
		P.from[i] = from ? from[i] : HessLevelSource{nullptr, 0, 0, 1};
		P.srcBorder[i] = 0;
		if (P.from[i].src) {
			if (skip % 2 != 0) return bhip_fail(ctx, BHIP_ERR_INVALID, "a shared level needs an octave at half the step");
			P.srcBorder[i] = makeLevel(sizes[i], skip / 2).border;
		}
	}
	P.srcW = skip >= 2 ? ii.width / (skip / 2) : 0;
	P.srcH = skip >= 2 ? ii.height / (skip / 2) : 0;
	if (P.w <= 0 || P.h <= 0) return BHIP_OK;
	dim3 grid((P.w + 255) / 256, P.h, batch * nlevels);
	{
		// algorithmic bytes: the integral image once + every level's intensity written once
		const double bytes = 4.0 * ii.width * ii.height * batch + 4.0 * nlevels * (double)P.w * P.h * batch;
		ProfScope ps(ctx, skip == 1 ? "k_hessian_skip1" : "k_hessian_skipN", bytes);
		if (intTaps) hipLaunchKernelGGL(k_hessian<int>, grid, dim3(256), 0, ctx->stream, P);
		else hipLaunchKernelGGL(k_hessian<float>, grid, dim3(256), 0, ctx->stream, P);
	}
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}
