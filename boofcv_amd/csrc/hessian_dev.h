// Device-side Fast-Hessian response of ONE pixel of one level (shared by k_hessian and the on-demand outer levels of k_nms_scalespace).
// Reference: ImplIntegralImageFeatureIntensity.hessianInner / hessianBorder (see hessian.hip).
#pragma once
#include "common.h"

struct HessLevel {
	int size;
	int bS, bL, rF, rS;     // blockSmall, blockLarge, radiusFeature, radiusSkinny
	int border, lost;       // border (in output pixels), lostPixel
	float norm;
	// kernelDerivXX / YY / XY parameters for the border path
	int r1, r2, r3, b;
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

// det(Hessian) of output pixel (x, y) of a level with geometry L on the octave lattice `skip` (w x h outputs): hessianInner inside the
// level's border, the clamped border form outside.  d = integral image of this frame.
template <class T>
__device__ __forceinline__ float hessianCompute(const T* __restrict__ d, int stride, int W, int H, const HessLevel& L, int skip, int w, int h, int x, int y) {
	const bool inner = x >= L.border && x < w - L.border && y >= L.border && y < h - L.border;
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
	return Dxx * Dyy - 0.81f * Dxy * Dxy;
}

static inline HessLevel bhipMakeHessLevel(int size, int skip) {
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

