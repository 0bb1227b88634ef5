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
#include "hessian_dev.h"

struct HessParams {
	ImgView ii;
	int skip, w, h, nlevels;
	int nrun, runLevel[BHIP_MAX_LEVELS];   // the levels this launch produces (outer levels may be left to k_nms_scalespace)
	float* out;             // [image][level][h][outStride]
	long long levelStride, imageStrideOut;
	int outStride;
	HessLevel lv[BHIP_MAX_LEVELS];
	HessLevelSource from[BHIP_MAX_LEVELS];
	int srcBorder[BHIP_MAX_LEVELS];   // shared levels: the producing octave's (step skip/2) inner-region border, width and height
	int srcW, srcH;
};

template <class T>
__global__ __launch_bounds__(256) void k_hessian(HessParams P) {
	const int x = blockIdx.x * blockDim.x + threadIdx.x;
	const int y = blockIdx.y;
	const int img = blockIdx.z / P.nrun;
	const int level = P.runLevel[blockIdx.z - img * P.nrun];
	if (x >= P.w) return;
	const HessLevel L = P.lv[level];
	if (P.from[level].src) {
		// Same kernel size one octave down.  A response depends on the pixel and the kernel size, and on which of the reference's two
		// forms evaluates it: hessianInner sums the Dyy boxes as ((br - bl) - tr) + tl, the border form (block_zero) as ((br - tr) - bl) + tl,
		// so the two can differ in the last bit.  The inner regions of the two octaves are not the same set of pixels (borderOrig depends
		// on the step), hence: copy where both octaves use the same form, compute in place on the few rows / columns where they do not.
		const bool inner = x >= L.border && x < P.w - L.border && y >= L.border && y < P.h - L.border;
		const int bp = P.srcBorder[level];
		const int px = 2 * x, py = 2 * y;
		const bool srcInner = px >= bp && px < P.srcW - bp && py >= bp && py < P.srcH - bp;
		if (srcInner == inner) {
			const HessLevelSource S = P.from[level];
			if (S.inPlace) return;   // the producing octave has written this pixel into this very plane
			P.out[(long long)img * P.imageStrideOut + (long long)level * P.levelStride + (long long)y * P.outStride + x] =
				S.src[(long long)img * S.imageStride + (long long)(y * S.step) * S.stride + x * S.step];
			return;
		}
	}
	const T* __restrict__ d = (const T*)P.ii.data + (long long)img * P.ii.imageStride;
	P.out[(long long)img * P.imageStrideOut + (long long)level * P.levelStride + (long long)y * P.outStride + x] =
		hessianCompute<T>(d, P.ii.stride, P.ii.width, P.ii.height, L, P.skip, P.w, P.h, x, y);
}

int bhip_launch_hessian(bhip_ctx* ctx, ImgView ii, int batch, int skip, int nlevels, const int* sizes, float* intensity, long long levelStride,
						long long imageStrideOut, int outStride, const HessLevelSource* from, bool intTaps, unsigned int skipMask) {
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
		P.lv[i] = bhipMakeHessLevel(sizes[i], skip);
		P.from[i] = from ? from[i] : HessLevelSource{nullptr, 0, 0, 1, 0};
		P.srcBorder[i] = 0;
		if (P.from[i].src) {
			if (skip % 2 != 0) return bhip_fail(ctx, BHIP_ERR_INVALID, "a shared level needs an octave at half the step");
			P.srcBorder[i] = bhipMakeHessLevel(sizes[i], skip / 2).border;
		}
	}
	P.srcW = skip >= 2 ? ii.width / (skip / 2) : 0;
	P.srcH = skip >= 2 ? ii.height / (skip / 2) : 0;
	P.nrun = 0;
	for (int i = 0; i < nlevels; i++)
		if (!(skipMask & (1u << i))) P.runLevel[P.nrun++] = i;
	if (P.w <= 0 || P.h <= 0 || P.nrun == 0) return BHIP_OK;
	dim3 grid((P.w + 255) / 256, P.h, batch * P.nrun);
	{
		// algorithmic bytes: the integral image once + every level's intensity written once
		const double bytes = 4.0 * ii.width * ii.height * batch + 4.0 * P.nrun * (double)P.w * P.h * batch;
		ProfScope ps(ctx, skip == 1 ? "k_hessian_skip1" : "k_hessian_skipN", bytes);
		if (intTaps) hipLaunchKernelGGL(k_hessian<int>, grid, dim3(256), 0, ctx->stream, P);
		else hipLaunchKernelGGL(k_hessian<float>, grid, dim3(256), 0, ctx->stream, P);
	}
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}
