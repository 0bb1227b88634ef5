// K2+K3 fused: one octave of the Fast-Hessian detector per launch, LDS tiled.  The four intensity levels of a tile are built in LDS from
// an LDS copy of the integral-image patch and consumed there by the non-maximum / scale-space test -- the intensity images never
// reach HBM.  Arithmetic and decisions are the same as the stand-alone kernels (hessian.hip, detect.hip), so key points stay bit-exact.
//
// Reference: IntegralImageFeatureIntensity.hessian (F:alg/feature/detect/intensity/IntegralImageFeatureIntensity.java:43-56,
//            impl/ImplIntegralImageFeatureIntensity.java:72-213), NonMaxBlock / NonMaxBlockSearchStrict.Max
//            (F:alg/feature/detect/extract/NonMaxBlock.java:69-94, NonMaxBlockSearchStrict.java:56-79,196-221),
//            FastHessianFeatureDetector.detectOctave / findLocalScaleSpaceMax / checkMax / polyPeak
//            (F:alg/feature/detect/interest/FastHessianFeatureDetector.java:198-350).
//
// Tile: TX x TY output pixels plus a halo of `radius` (the NMS neighbourhood; the 3x3 scale-space test needs 1), ITW = TX + 2*halo is a
// power of two so an item index splits into (row, column) with shifts.  ii patch = every tap any inner pixel of the tile can touch:
// [(x0-halo)*skip - rFmax - 1, (x0+TX-1+halo)*skip + rFmax].  Pixels that need the reference's clamped border formula (hessianBorder)
// read the integral image from global memory; they only exist in tiles along the image frame.
// Bound: HBM by the algorithmic count (ii read once per octave, nothing written but key points); in practice LDS-read / issue bound:
// 32 taps per intensity value.
#include "common.h"
#include <cfloat>

struct FusedLevel {
	int size;
	int bS, bL, rF, rS;
	int border, lost;
	float norm;
	int r1, r2, r3, b;
};
struct FusedMid {
	int level;            // index of the mid level
	int border;           // ignoreBorder = size/(2*skip)
	int nbx;              // NMS blocks per row
	unsigned int bitBase;
	int sizeMid, sizeLower;
};
struct FusedParams {
	ImgView ii;
	int skip, w, h, nlevels;
	FusedLevel lv[BHIP_MAX_LEVELS];
	int TX, TY, HR, ITW, ITWlog, ITH, ITp;   // tile geometry (intensity space); ITp = LDS pitch of an intensity row
	int IW, IH, IWp, rFmax;                   // ii patch geometry
	int radius;
	float threshold;
	int nmid;
	FusedMid mid[BHIP_MAX_LEVELS];
	unsigned int* bitmap;
	int bitmapWords;
	KeyPoint* cand;
	int* candCount;
	int cap;
	int batch;
	// Levels whose kernel size appears again in the next octave (size 27 and 51 of octave 1 are levels 0 and 1 of octave 2, ...): the
	// intensity of a pixel does not depend on the sampling step, so every second pixel of those levels is written out in the next
	// octave's layout [image][slot][expH][expW] and the next octave copies instead of recomputing (hessian.hip).
	int nexp;
	int expLevel[2];
	float* expOut;
	int expW, expH;
	long long expImageStride;
	long long expSlotOff[2];
	int ablate;   // timing experiments only (BHIP_FUSED_ABLATE): 1 skip the intensity phase, 2 skip the NMS phase, 4 skip staging
};

__device__ __forceinline__ float fblock_zero(const float* __restrict__ d, int stride, int W, int H, int x0, int y0, int x1, int y1) {
	x0 = min(x0, W - 1); y0 = min(y0, H - 1); x1 = min(x1, W - 1); y1 = min(y1, H - 1);
	float br = 0, tr = 0, bl = 0, tl = 0;
	if (x1 >= 0 && y1 >= 0) br = d[(long long)y1 * stride + x1];
	if (y0 >= 0 && x1 >= 0) tr = d[(long long)y0 * stride + x1];
	if (x0 >= 0 && y1 >= 0) bl = d[(long long)y1 * stride + x0];
	if (x0 >= 0 && y0 >= 0) tl = d[(long long)y0 * stride + x0];
	return br - tr - bl + tl;
}
// block_zero on the staged patch: every clamped corner of a pixel of this tile lies inside the patch (clamping only moves a corner
// towards the pixel), so the reference's border formula needs no global memory either.
__device__ __forceinline__ float lblock_zero(const float* iiT, int pitch, int X0, int Y0, int W, int H, int x0, int y0, int x1, int y1) {
	x0 = min(x0, W - 1); y0 = min(y0, H - 1); x1 = min(x1, W - 1); y1 = min(y1, H - 1);
	const int cx0 = max(x0, 0) - X0, cy0 = max(y0, 0) - Y0, cx1 = max(x1, 0) - X0, cy1 = max(y1, 0) - Y0;
	const float vbr = iiT[cy1 * pitch + cx1], vtr = iiT[cy0 * pitch + cx1], vbl = iiT[cy1 * pitch + cx0], vtl = iiT[cy0 * pitch + cx0];
	const float br = (x1 >= 0 && y1 >= 0) ? vbr : 0.0f;
	const float tr = (y0 >= 0 && x1 >= 0) ? vtr : 0.0f;
	const float bl = (x0 >= 0 && y1 >= 0) ? vbl : 0.0f;
	const float tl = (x0 >= 0 && y0 >= 0) ? vtl : 0.0f;
	return br - tr - bl + tl;
}
__device__ __forceinline__ float fpolyPeak(float lower, float middle, float upper) {
	const float a = 0.5f * lower - middle + 0.5f * upper;
	const float b = 0.5f * upper - 0.5f * lower;
	if (a == 0.0f) return 0.0f;
	return -b / (2.0f * a);
}

__global__ __launch_bounds__(256) void k_detect_fused(FusedParams P) {
	extern __shared__ __attribute__((aligned(16))) float fl[];
	float* iiT = fl;                                  // [IH][IWp]
	float* inten = fl + (size_t)P.IH * P.IWp;         // [nlevels][ITH][ITp]
	const int tid = threadIdx.x;
	const int img = blockIdx.z;
	const int x0 = blockIdx.x * P.TX, y0 = blockIdx.y * P.TY;
	const int s = P.skip;
	const float* __restrict__ d = P.ii.data + (long long)img * P.ii.imageStride;
	const int stride = P.ii.stride, W = P.ii.width, H = P.ii.height;
	const int X0 = (x0 - P.HR) * s - P.rFmax - 1, Y0 = (y0 - P.HR) * s - P.rFmax - 1;

	// ---- stage the integral-image patch (rows of IW floats, coalesced)
	if (!BHIP_ABLATE(P, 4)) {
		const int tx = tid & 63, ty = tid >> 6;
		for (int ry = ty; ry < P.IH; ry += 4) {
			const int gy = Y0 + ry;
			const bool rowOk = gy >= 0 && gy < H;
			const float* __restrict__ src = d + (long long)(rowOk ? gy : 0) * stride;
			for (int rx = tx; rx < P.IW; rx += 64) {
				const int gx = X0 + rx;
				iiT[ry * P.IWp + rx] = (rowOk && gx >= 0 && gx < W) ? src[gx] : 0.0f;
			}
		}
	}
	__syncthreads();

	// ---- intensity of every level over the tile + halo
	const int itemsPerLevel = P.ITH << P.ITWlog;
	for (int L = 0; L < (BHIP_ABLATE(P, 1) ? 0 : P.nlevels); L++) {
		const FusedLevel V = P.lv[L];
		float* out = inten + (size_t)L * P.ITH * P.ITp;
		// tap offsets inside the patch, relative to (xx - X0, yy - Y0)
		const int bS = V.bS;
		const int cxx = -V.rF - 1, rxx = -V.rS - 1;           // Dxx: first column / top row
		const int cyy = -V.rS - 1, ryy = -V.rF - 1;           // Dyy: left column / first row
		const int cxy = -bS - 1, rxy = -bS - 1;               // Dxy
		for (int it = tid; it < itemsPerLevel; it += 256) {
			const int px = it & (P.ITW - 1), py = it >> P.ITWlog;
			const int x = x0 - P.HR + px, y = y0 - P.HR + py;
			float det;
			if (x < 0 || x >= P.w || y < 0 || y >= P.h) {
				det = -INFINITY;  // outside the image: never >= anything, as if the neighbourhood were clamped
			} else {
				const int xx = x * s, yy = y * s;
				float Dxx, Dyy, Dxy;
				const bool inner = x >= V.border && x < P.w - V.border && y >= V.border && y < P.h - V.border;
				if (inner) {
					const int lx = xx - X0, ly = yy - Y0;
					const int pitch = P.IWp;
					{
						const float* t = iiT + (ly + rxx) * pitch + lx + cxx;
						const float* bt = t + V.bL * pitch;
						Dxx = bt[3 * bS] - t[3 * bS] - bt[0] + t[0];
						Dxx -= 3 * (bt[2 * bS] - t[2 * bS] - bt[bS] + t[bS]);
					}
					{
						const float* l = iiT + (ly + ryy) * pitch + lx + cyy;
						const float* r = l + V.bL;
						const int ro1 = bS * pitch;
						Dyy = r[3 * ro1] - l[3 * ro1] - r[0] + l[0];
						Dyy -= 3 * (r[2 * ro1] - l[2 * ro1] - r[ro1] + l[ro1]);
					}
					{
						const float* y1 = iiT + (ly + rxy) * pitch + lx + cxy;
						const float* y2 = y1 + bS * pitch;
						const float* y3 = y2 + pitch;
						const float* y4 = y3 + bS * pitch;
						const int x3 = bS + 1, x4 = x3 + bS;
						Dxy = y2[bS] - y1[bS] - y2[0] + y1[0];
						Dxy -= y2[x4] - y1[x4] - y2[x3] + y1[x3];
						Dxy += y4[x4] - y3[x4] - y4[x3] + y3[x3];
						Dxy -= y4[bS] - y3[bS] - y4[0] + y3[0];
					}
				} else {
					const int pitch = P.IWp;
					float ret = 0;
					ret += lblock_zero(iiT, pitch, X0, Y0, W, H, xx - V.r2 - 1, yy - V.r3 - 1, xx + V.r2, yy + V.r3) * 1.0f;
					ret += lblock_zero(iiT, pitch, X0, Y0, W, H, xx - V.r1 - 1, yy - V.r3 - 1, xx + V.r1, yy + V.r3) * -3.0f;
					Dxx = ret;
					ret = 0;
					ret += lblock_zero(iiT, pitch, X0, Y0, W, H, xx - V.r3 - 1, yy - V.r2 - 1, xx + V.r3, yy + V.r2) * 1.0f;
					ret += lblock_zero(iiT, pitch, X0, Y0, W, H, xx - V.r3 - 1, yy - V.r1 - 1, xx + V.r3, yy + V.r1) * -3.0f;
					Dyy = ret;
					ret = 0;
					const int b = V.b;
					ret += lblock_zero(iiT, pitch, X0, Y0, W, H, xx - b - 1, yy - b - 1, xx - 1, yy - 1) * 1.0f;
					ret += lblock_zero(iiT, pitch, X0, Y0, W, H, xx, yy - b - 1, xx + b, yy - 1) * -1.0f;
					ret += lblock_zero(iiT, pitch, X0, Y0, W, H, xx, yy, xx + b, yy + b) * 1.0f;
					ret += lblock_zero(iiT, pitch, X0, Y0, W, H, xx - b - 1, yy, xx - 1, yy + b) * -1.0f;
					Dxy = ret;
				}
				Dxx *= V.norm;
				Dxy *= V.norm;
				Dyy *= V.norm;
				det = Dxx * Dyy - 0.81f * Dxy * Dxy;
			}
			out[py * P.ITp + px] = det;
		}
	}
	__syncthreads();

	// ---- strict (2r+1)^2 maximum + scale-space test on the mid levels, from LDS
	const int r = P.radius;
	const int wave = tid >> 6, lane = tid & 63;
	const int rows = BHIP_ABLATE(P, 2) ? 0 : P.nmid * P.TY;
	for (int row = wave; row < rows; row += 4) {
		const int m = row / P.TY;
		const int py = row - m * P.TY;
		for (int px = lane; px < P.TX; px += 64) {
			const FusedMid M = P.mid[m];
			const int x = x0 + px, y = y0 + py;
			const int b = M.border;
			if (x < b || x >= P.w - b || y < b || y >= P.h - b) continue;
			const float* mid = inten + (size_t)M.level * P.ITH * P.ITp + (py + P.HR) * P.ITp + (px + P.HR);
			const float val = mid[0];
			if (!(val >= P.threshold) || val == FLT_MAX) continue;
			bool isMax = true;
			for (int j = -r; j <= r && isMax; j++)
				for (int i = -r; i <= r; i++) {
					if ((i | j) != 0 && mid[j * P.ITp + i] >= val) { isMax = false; break; }
				}
			if (!isMax) continue;
			const int ignoreR = b + r;
			if (x < ignoreR || x >= P.w - ignoreR || y < ignoreR || y >= P.h - ignoreR) continue;
			const float* lower = mid - (size_t)P.ITH * P.ITp;
			const float* upper = mid + (size_t)P.ITH * P.ITp;
			bool ok = true;
			for (int j = -1; j <= 1 && ok; j++)
				for (int i = -1; i <= 1; i++) {
					if (lower[j * P.ITp + i] >= val || upper[j * P.ITp + i] >= val) { ok = false; break; }
				}
			if (!ok) continue;
			const float peakX = fpolyPeak(mid[-1], val, mid[1]);
			const float peakY = fpolyPeak(mid[-P.ITp], val, mid[P.ITp]);
			const float peakS = fpolyPeak(lower[0], val, upper[0]);
			const float interpX = ((float)x + peakX) * (float)s;
			const float interpY = ((float)y + peakY) * (float)s;
			const float interpS = (float)M.sizeMid + peakS * (float)(M.sizeMid - M.sizeLower);
			const double scale = 1.2 * (double)interpS / 9.0;
			const int step = r + 1;
			const unsigned int bit = M.bitBase + (unsigned)((y - b) / step) * (unsigned)M.nbx + (unsigned)((x - b) / step);
			atomicOr(&P.bitmap[(long long)img * P.bitmapWords + (bit >> 5)], 1u << (bit & 31));
			const int slot = atomicAdd(&P.candCount[img], 1);
			if (slot < P.cap) {
				KeyPoint kp;
				kp.x = (double)interpX;
				kp.y = (double)interpY;
				kp.scale = scale;
				kp.key = bit;
				kp.pad = 0;
				P.cand[(long long)img * P.cap + slot] = kp;
			}
		}
	}
}


// ---------------------------------------------------------------------------------------------------------------
// Compile-time geometry variant for the reference's default schedule (sizes SIZE0 + i*STEPSZ, NL levels, NMS radius R): every tap
// offset becomes an immediate of a ds_read, pairs of taps in one row fuse into ds_read2, and the address arithmetic per intensity value
// shrinks to one base.  Same arithmetic, same decisions as k_detect_fused.
template <int SKIP, int SIZE0, int STEPSZ, int NL, int R, int ITWT, int TYT, int NTT = 256>
struct FixedGeo {
	static constexpr int NT = NTT, NW = NTT / 64;   // threads / waves per workgroup
	static constexpr int TX = ITWT - 2 * R, TY = TYT, ITW = ITWT, ITH = TYT + 2 * R, ITp = ITWT + 1, HALO = R;
	static constexpr int sizeMax = SIZE0 + (NL - 1) * STEPSZ;
	static constexpr int rFmax = sizeMax / 2;
	static constexpr int IW = (TX - 1 + 2 * R) * SKIP + 2 * rFmax + 2, IH = (TY - 1 + 2 * R) * SKIP + 2 * rFmax + 2;
	// The patch is stored as SKIP column-phase planes: element (row, col) lives at plane (col % SKIP), row, col / SKIP.  A tap of the
	// lanes' consecutive output pixels (columns xx = x*SKIP apart) then reads consecutive LDS words -- no bank conflicts for SKIP > 1.
	static constexpr int IWp = ((IW + SKIP - 1) / SKIP) | 1;      // pitch of one plane row
	static constexpr int plane = IH * IWp;
	// LDS: the patch, the two dense mid levels (1 and 2 of the four), survivor list + their nine sparse outer-level values
	static constexpr int SURV = 2 * ((TX + R) / (R + 1)) * ((TY + R) / (R + 1));   // strict (2R+1)^2 maxima of two mid levels: one per (R+1)^2 block at most
	static constexpr int CHUNK = 64;   // survivors whose nine outer-level values are held at a time
	static constexpr int NCAND = TX * TY + 2;                     // two mid levels, at most every second core pixel (rounded up) each
	static_assert(TX < 64 && TY < 64, "16-bit NMS candidate codes");
	static constexpr int SPARSE = CHUNK * 9 > (NCAND + 1) / 2 ? CHUNK * 9 : (NCAND + 1) / 2;   // floats: sparse values, before that the candidate codes
	static constexpr int ldsFloats = SKIP * plane + 2 * ITH * ITp + SURV + SPARSE + 4;
	static constexpr int K0 = rFmax + 1;                          // column of (xx) inside the patch, minus SKIP*(x - x0 + R)
	// LDS offset of the tap at (row offset ro, column offset co) relative to the pixel's base pointer (row yy-Y0, column slot x-x0+R)
	static constexpr int tap(int ro, int co) { return ((K0 + co) % SKIP) * plane + ro * IWp + (K0 + co) / SKIP; }
	// run-time form for the border formula: absolute patch coordinates (r, c) -> LDS index
	static __device__ __forceinline__ int at(int r, int c) { return (c % SKIP) * plane + r * IWp + c / SKIP; }
	static constexpr int size(int L) { return SIZE0 + L * STEPSZ; }
	static constexpr int bS(int L) { return size(L) / 3; }
	static constexpr int bL(int L) { return size(L) - bS(L) - 1; }
	static constexpr int rF(int L) { return size(L) / 2; }
	static constexpr int rS(int L) { return bL(L) / 2; }
	static constexpr int border(int L) { return (rF(L) + 1 + (SKIP - (rF(L) + 1) % SKIP)) / SKIP; }
};

// block_zero on the phase-plane patch (see lblock_zero)
// T = float (GrayF32 integral image) or int (GrayS32: the box sums are exact integers and become floats exactly where the reference's S32
// code assigns them to a float, ImplIntegralImageFeatureIntensity.java:245-390)
template <class T> struct TapVec2;
template <> struct TapVec2<float> { typedef float2 type; };
template <> struct TapVec2<int> { typedef int2 type; };

template <class G, class T>
__device__ __forceinline__ T pblock_zero(const T* iiT, int X0, int Y0, int W, int H, int x0, int y0, int x1, int y1) {
	x0 = min(x0, W - 1); y0 = min(y0, H - 1); x1 = min(x1, W - 1); y1 = min(y1, H - 1);
	const int cx0 = max(x0, 0) - X0, cy0 = max(y0, 0) - Y0, cx1 = max(x1, 0) - X0, cy1 = max(y1, 0) - Y0;
	const T vbr = iiT[G::at(cy1, cx1)], vtr = iiT[G::at(cy0, cx1)], vbl = iiT[G::at(cy1, cx0)], vtl = iiT[G::at(cy0, cx0)];
	const T br = (x1 >= 0 && y1 >= 0) ? vbr : T(0);
	const T tr = (y0 >= 0 && x1 >= 0) ? vtr : T(0);
	const T bl = (x0 >= 0 && y1 >= 0) ? vbl : T(0);
	const T tl = (x0 >= 0 && y0 >= 0) ? vtl : T(0);
	return br - tr - bl + tl;
}

// the 32 taps of one inner pixel, reference order (hessianInner :183-201); c = patch row of yy, column slot of x
template <class G, int L, class T>
__device__ __forceinline__ float fusedInnerDet(const T* c) {
	constexpr int size = G::size(L), bS = G::bS(L), bLg = G::bL(L), rF = G::rF(L), rS = G::rS(L);
	constexpr float norm = 1.0f / (float)(size * size);
#define TAP(ro, co) c[G::tap((ro), (co))]
	float Dxx, Dyy, Dxy;
	{
		constexpr int rt = -rS - 1, rb = rt + bLg, c0 = -rF - 1;
		Dxx = (float)(TAP(rb, c0 + 3 * bS) - TAP(rt, c0 + 3 * bS) - TAP(rb, c0) + TAP(rt, c0));
		Dxx -= (float)(T(3) * (TAP(rb, c0 + 2 * bS) - TAP(rt, c0 + 2 * bS) - TAP(rb, c0 + bS) + TAP(rt, c0 + bS)));
	}
	{
		constexpr int r0 = -rF - 1, cl = -rS - 1, cr = cl + bLg;
		Dyy = (float)(TAP(r0 + 3 * bS, cr) - TAP(r0 + 3 * bS, cl) - TAP(r0, cr) + TAP(r0, cl));
		Dyy -= (float)(T(3) * (TAP(r0 + 2 * bS, cr) - TAP(r0 + 2 * bS, cl) - TAP(r0 + bS, cr) + TAP(r0 + bS, cl)));
	}
	{
		constexpr int ry1 = -bS - 1, ry2 = ry1 + bS, ry3 = ry2 + 1, ry4 = ry3 + bS, c0 = -bS - 1;
		constexpr int x3 = bS + 1, x4 = x3 + bS;
		Dxy = (float)(TAP(ry2, c0 + bS) - TAP(ry1, c0 + bS) - TAP(ry2, c0) + TAP(ry1, c0));
		Dxy -= (float)(TAP(ry2, c0 + x4) - TAP(ry1, c0 + x4) - TAP(ry2, c0 + x3) + TAP(ry1, c0 + x3));
		Dxy += (float)(TAP(ry4, c0 + x4) - TAP(ry3, c0 + x4) - TAP(ry4, c0 + x3) + TAP(ry3, c0 + x3));
		Dxy -= (float)(TAP(ry4, c0 + bS) - TAP(ry3, c0 + bS) - TAP(ry4, c0) + TAP(ry3, c0));
	}
#undef TAP
	Dxx *= norm;
	Dxy *= norm;
	Dyy *= norm;
	return Dxx * Dyy - 0.81f * Dxy * Dxy;
}

// Two horizontally adjacent pixels per call (float taps only): their taps are adjacent LDS words, so every pair is one ds_read2 and
// the box arithmetic runs on packed fp32 (v_pk_add_f32 / v_pk_mul_f32) -- component-wise the same operations in the same order as
// fusedInnerDet, at about half the instructions.  The kernel is bound by instruction issue (a wave64 instruction holds its SIMD for 4 cycles).
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <class G, int L>
__device__ __forceinline__ f32x2 fusedInnerDet2(const float* c) {
	constexpr int size = G::size(L), bS = G::bS(L), bLg = G::bL(L), rF = G::rF(L), rS = G::rS(L);
	constexpr float norm = 1.0f / (float)(size * size);
#define TAP2(ro, co) (f32x2{c[G::tap((ro), (co))], c[G::tap((ro), (co)) + 1]})
	f32x2 Dxx, Dyy, Dxy;
	{
		constexpr int rt = -rS - 1, rb = rt + bLg, c0 = -rF - 1;
		Dxx = TAP2(rb, c0 + 3 * bS) - TAP2(rt, c0 + 3 * bS) - TAP2(rb, c0) + TAP2(rt, c0);
		Dxx -= 3.0f * (TAP2(rb, c0 + 2 * bS) - TAP2(rt, c0 + 2 * bS) - TAP2(rb, c0 + bS) + TAP2(rt, c0 + bS));
	}
	{
		constexpr int r0 = -rF - 1, cl = -rS - 1, cr = cl + bLg;
		Dyy = TAP2(r0 + 3 * bS, cr) - TAP2(r0 + 3 * bS, cl) - TAP2(r0, cr) + TAP2(r0, cl);
		Dyy -= 3.0f * (TAP2(r0 + 2 * bS, cr) - TAP2(r0 + 2 * bS, cl) - TAP2(r0 + bS, cr) + TAP2(r0 + bS, cl));
	}
	{
		constexpr int ry1 = -bS - 1, ry2 = ry1 + bS, ry3 = ry2 + 1, ry4 = ry3 + bS, c0 = -bS - 1;
		constexpr int x3 = bS + 1, x4 = x3 + bS;
		Dxy = TAP2(ry2, c0 + bS) - TAP2(ry1, c0 + bS) - TAP2(ry2, c0) + TAP2(ry1, c0);
		Dxy -= TAP2(ry2, c0 + x4) - TAP2(ry1, c0 + x4) - TAP2(ry2, c0 + x3) + TAP2(ry1, c0 + x3);
		Dxy += TAP2(ry4, c0 + x4) - TAP2(ry3, c0 + x4) - TAP2(ry4, c0 + x3) + TAP2(ry3, c0 + x3);
		Dxy -= TAP2(ry4, c0 + bS) - TAP2(ry3, c0 + bS) - TAP2(ry4, c0) + TAP2(ry3, c0);
	}
#undef TAP2
	Dxx *= norm;
	Dxy *= norm;
	Dyy *= norm;
	return Dxx * Dyy - 0.81f * Dxy * Dxy;
}

// Intensity of level L at intensity-space pixel (x, y), any position class: -inf outside the image (never >= anything, as if the
// neighbourhood were clamped), the unrolled inner form (hessianInner) for inner pixels, the clamped border form (hessianBorder) otherwise
template <class G, int SKIP, int L, class T>
__device__ __forceinline__ float fusedPixel(int W, int H, int pw, int ph, const T* iiT, int x, int y, int x0, int y0, int X0, int Y0) {
	constexpr int size = G::size(L), bS = G::bS(L), bLg = G::bL(L), border = G::border(L), R = G::HALO;
	constexpr float norm = 1.0f / (float)(size * size);
	constexpr int r1 = bS / 2, r2 = bS + r1, r3 = bLg / 2, b = bS;
	if (x < 0 || x >= pw || y < 0 || y >= ph) return -INFINITY;
	const int xx = x * SKIP, yy = y * SKIP;
	const bool inner = x >= border && x < pw - border && y >= border && y < ph - border;
	if (inner) return fusedInnerDet<G, L, T>(iiT + (yy - Y0) * G::IWp + (x - x0 + R));
	float Dxx, Dyy, Dxy;
	T ret = 0;
	ret += pblock_zero<G, T>(iiT, X0, Y0, W, H, xx - r2 - 1, yy - r3 - 1, xx + r2, yy + r3) * T(1);
	ret += pblock_zero<G, T>(iiT, X0, Y0, W, H, xx - r1 - 1, yy - r3 - 1, xx + r1, yy + r3) * T(-3);
	Dxx = (float)ret;
	ret = 0;
	ret += pblock_zero<G, T>(iiT, X0, Y0, W, H, xx - r3 - 1, yy - r2 - 1, xx + r3, yy + r2) * T(1);
	ret += pblock_zero<G, T>(iiT, X0, Y0, W, H, xx - r3 - 1, yy - r1 - 1, xx + r3, yy + r1) * T(-3);
	Dyy = (float)ret;
	ret = 0;
	ret += pblock_zero<G, T>(iiT, X0, Y0, W, H, xx - b - 1, yy - b - 1, xx - 1, yy - 1) * T(1);
	ret += pblock_zero<G, T>(iiT, X0, Y0, W, H, xx, yy - b - 1, xx + b, yy - 1) * T(-1);
	ret += pblock_zero<G, T>(iiT, X0, Y0, W, H, xx, yy, xx + b, yy + b) * T(1);
	ret += pblock_zero<G, T>(iiT, X0, Y0, W, H, xx - b - 1, yy, xx - 1, yy + b) * T(-1);
	Dxy = (float)ret;
	Dxx *= norm;
	Dxy *= norm;
	Dyy *= norm;
	return Dxx * Dyy - 0.81f * Dxy * Dxy;
}

// Out-of-line form for the rare callers (frame tiles, NMS survivors, exported outer levels): one copy of the 32-tap body + the border
// formula per level instead of one per call site keeps the kernel's register count at what the dense loop needs.
template <class G, int SKIP, int L, class T>
__device__ __noinline__ float fusedPixelCall(int W, int H, int pw, int ph, const T* iiT, int x, int y, int x0, int y0, int X0, int Y0) {
	return fusedPixel<G, SKIP, L, T>(W, H, pw, ph, iiT, x, y, x0, y0, X0, Y0);
}

// Dense intensity tile (core + halo) of level L into `out` ([ITH][ITp])
template <class G, int SKIP, int R, int L, class T>
__device__ __forceinline__ void fusedLevelDense(const FusedParams& P, const T* iiT, float* out, int tid, int x0, int y0, int X0, int Y0) {
	constexpr int border = G::border(L);
	constexpr int pitch = G::IWp;
	// tile + halo entirely made of inner pixels of this level (true for all but the tiles along the image frame): no per-pixel tests
	const bool interior = x0 - R >= border && x0 + G::TX + R <= P.w - border && y0 - R >= border && y0 + G::TY + R <= P.h - border;
	if (interior) {
		const int rowBase = (y0 - R) * SKIP - Y0;   // == rFmax + 1
		if constexpr (SKIP == 1 && T(0.5f) != T(0)) {   // float taps, octave 0: two pixels per thread on packed fp32 (no gain on the staging-bound octave 1)
			for (int it = tid; it < G::ITH * (G::ITW / 2); it += G::NT) {
				const int px = 2 * (it & (G::ITW / 2 - 1)), py = it / (G::ITW / 2);
				const f32x2 det = fusedInnerDet2<G, L>((const float*)iiT + (rowBase + py * SKIP) * pitch + px);
				out[py * G::ITp + px] = det.x;
				out[py * G::ITp + px + 1] = det.y;
			}
		} else {
#pragma unroll 2
			for (int it = tid; it < G::ITH * G::ITW; it += G::NT) {
				const int px = it & (G::ITW - 1), py = it / G::ITW;
				const T* c = iiT + (rowBase + py * SKIP) * pitch + px;
				out[py * G::ITp + px] = fusedInnerDet<G, L, T>(c);
			}
		}
	} else {
#pragma unroll 1
		for (int it = tid; it < G::ITH * G::ITW; it += G::NT) {
			const int px = it & (G::ITW - 1), py = it / G::ITW;
			out[py * G::ITp + px] = fusedPixelCall<G, SKIP, L, T>(P.ii.width, P.ii.height, P.w, P.h, iiT, x0 - R + px, y0 - R + py, x0, y0, X0, Y0);
		}
	}
}

// checkMax on the lower and the upper level + polyPeak fits + emission for an NMS survivor of mid slot m at tile pixel (px, py):
// FastHessianFeatureDetector.findLocalScaleSpaceMax :278-297.  The outer level on the far side of the mid level (level 0 below mid level 1,
// level 3 above mid level 2) arrives as "all nine neighbours below val" + its centre value; the other neighbour level is the other dense
// mid level.
template <class G, int SKIP, int R>
__device__ __forceinline__ void fusedFinish(const FusedParams& P, const FusedMid& M, const float* mid1, const float* mid2, bool outerBelow, float outerCentre,
											int px, int py, int x, int y, float val, int img) {
	const bool lowMid = M.level == 1;
	const float* self = (lowMid ? mid1 : mid2) + (py + R) * G::ITp + (px + R);
	const float* other = (lowMid ? mid2 : mid1) + (py + R) * G::ITp + (px + R);
	if (!outerBelow) return;
	float dn[9];
	bool ok = true;
#pragma unroll
	for (int q = 0; q < 9; q++) {
		dn[q] = other[(q / 3 - 1) * G::ITp + (q % 3 - 1)];
		if (dn[q] >= val) ok = false;
	}
	if (!ok) return;
	const float lowerC = lowMid ? outerCentre : dn[4], upperC = lowMid ? dn[4] : outerCentre;
	const float peakX = fpolyPeak(self[-1], val, self[1]);
	const float peakY = fpolyPeak(self[-G::ITp], val, self[G::ITp]);
	const float peakS = fpolyPeak(lowerC, val, upperC);
	const float interpX = ((float)x + peakX) * (float)SKIP;
	const float interpY = ((float)y + peakY) * (float)SKIP;
	const float interpS = (float)M.sizeMid + peakS * (float)(M.sizeMid - M.sizeLower);
	const double scale = 1.2 * (double)interpS / 9.0;
	constexpr int step = R + 1;
	if (BHIP_ABLATE(P, 7)) return;   // timing experiments run on garbage: never emit
	const int b = M.border;
	const unsigned int bit = M.bitBase + (unsigned)((y - b) / step) * (unsigned)M.nbx + (unsigned)((x - b) / step);
	atomicOr(&P.bitmap[(long long)img * P.bitmapWords + (bit >> 5)], 1u << (bit & 31));
	const int slot = atomicAdd(&P.candCount[img], 1);
	if (slot < P.cap) {
		KeyPoint kp;
		kp.x = (double)interpX;
		kp.y = (double)interpY;
		kp.scale = scale;
		kp.key = bit;
		kp.pad = 0;
		P.cand[(long long)img * P.cap + slot] = kp;
	}
}

template <class T, int SKIP, int SIZE0, int STEPSZ, int NL, int R, int ITWT, int TYT, int NTT = 256>
__global__ __launch_bounds__(NTT) void k_detect_fused_fixed(FusedParams P) {
	typedef FixedGeo<SKIP, SIZE0, STEPSZ, NL, R, ITWT, TYT, NTT> G;
	extern __shared__ __attribute__((aligned(16))) float fl[];
	T* iiT = (T*)fl;   // 32-bit words either way
	float* inten = fl + SKIP * G::plane;   // the two dense mid levels, then the survivor list and its sparse outer-level values
	const int tid = threadIdx.x;
	// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share one L2), so block b takes
	// tile (b % 8) * chunk + b / 8 -- every XCD walks its own contiguous run of tiles (row after row of one image) and the halo
	// re-reads of neighbouring tiles hit that XCD's 4 MB L2 instead of going out to the fabric.  Speed only, never correctness.
	int bx, by, img;
	{
		// (32-bit arithmetic: the launcher refuses more than 2^31 tiles; a 64-bit division here costs ~250 scalar instructions per wave)
		const unsigned int tilesX = (unsigned)((P.w + G::TX - 1) / G::TX), tilesY = (unsigned)((P.h + G::TY - 1) / G::TY);
		const unsigned int perImage = tilesX * tilesY;
		const unsigned int ntiles = perImage * (unsigned)P.batch;
		const unsigned int chunk = (ntiles + 7u) >> 3;
		const unsigned int lin = (blockIdx.x & 7u) * chunk + (blockIdx.x >> 3);
		if ((blockIdx.x >> 3) >= chunk || lin >= ntiles) return;
		img = (int)(lin / perImage);
		const unsigned int rem = lin - (unsigned)img * perImage;
		by = (int)(rem / tilesX);
		bx = (int)(rem - (unsigned)by * tilesX);
	}
	const int x0 = bx * G::TX, y0 = by * G::TY;
	const T* __restrict__ d = (const T*)P.ii.data + (long long)img * P.ii.imageStride;
	const int stride = P.ii.stride, W = P.ii.width, H = P.ii.height;
	const int X0 = (x0 - R) * SKIP - G::rFmax - 1, Y0 = (y0 - R) * SKIP - G::rFmax - 1;
	bool staged = false;
	// Interior tiles (all but those along the image frame): the patch lies inside the image, so its rows go from global memory straight
	// into LDS (global_load_lds_dword: lane l of an instruction delivers its dword to M0-base + 4 l; no VGPR round trip, no ds_write, no
	// per-element address arithmetic -- staging used to be a quarter of this kernel's vector instructions).  One instruction per patch
	// row and 64-column chunk (SKIP == 1) or per row and column phase (SKIP > 1: the lanes read every SKIP-th column, which lands
	// them contiguously in that phase's plane).  The barrier below waits for the loads (vmcnt) before anything reads the patch.
	{
#ifdef BHIP_EXPERIMENTS
		const bool dmaOff = BHIP_ABLATE(P, 16);
#else
		const bool dmaOff = false;
#endif
		// (measured: octave 0 1.10 -> 1.03 ms per 64 frames; on octave 1 the stride-2 dword form needs 276 instructions per tile and is
		// 6 % slower than the 16-byte register-staged form below, so only SKIP == 1 takes this path)
		const bool dmaOk = SKIP == 1 && !dmaOff && !BHIP_ABLATE(P, 4) && X0 >= 0 && Y0 >= 0 && X0 + G::IW <= W && Y0 + G::IH <= H;
		if (dmaOk) {
			const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
			const T* src0 = d + (long long)Y0 * stride + X0;
			constexpr int PCOLS = (G::IW + SKIP - 1) / SKIP;          // columns of one phase plane
			constexpr int CHUNKS = (PCOLS + 63) / 64;
			for (int ry = wave; ry < G::IH; ry += G::NW) {
				const T* srow = src0 + (long long)ry * stride;
#pragma unroll
				for (int ph = 0; ph < SKIP; ph++) {
#pragma unroll
					for (int cc = 0; cc < CHUNKS; cc++) {
						const int j = cc * 64 + lane;                   // column slot inside the plane
						if (j * SKIP + ph < G::IW)
							__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(srow + j * SKIP + ph),
															 (__attribute__((address_space(3))) void*)(iiT + ph * G::plane + ry * G::IWp + cc * 64), 4, 0, 0);
					}
				}
			}
			staged = true;
		}
	}
	// (fall-back forms below: frame tiles, and the earlier register-staged interior forms kept for cross-checks)
	// Interior tiles (all but those along the image frame): the patch lies inside the image and, with the tile pitch a multiple of four
	// columns, starts on (SKIP == 1) or two columns after (SKIP == 2) a 16-byte boundary -- whole rows are fetched with 16-byte loads,
	// no per-element guards.  One (row, 4-column group) item per thread and pass; all loads are issued before the first LDS store.
	if constexpr (SKIP == 1 || SKIP == 2) {
		constexpr int SH = SKIP == 1 ? 0 : 2;                  // patch column 0 sits SH elements after the aligned start
		constexpr int IW4 = (G::IW + SH + 3) / 4;
		constexpr int ITEMS = G::IH * IW4, NIT = (ITEMS + G::NT - 1) / G::NT;
		const T* src0 = d + (long long)Y0 * stride + (X0 - SH);
		const bool vecOk = !staged && !BHIP_ABLATE(P, 4) && X0 - SH >= 0 && Y0 >= 0 && X0 - SH + 4 * IW4 <= W && Y0 + G::IH <= H && (stride & 3) == 0 &&
						   (((unsigned long long)src0) & 15ull) == 0;
		if (vecOk) {
			int4 v[NIT];
#pragma unroll
			for (int it = 0; it < NIT; it++) {
				const int item = tid + G::NT * it;
				if (item < ITEMS) {
					const int row = item / IW4, c4 = item - row * IW4;
					v[it] = *(const int4*)(src0 + (long long)row * stride + 4 * c4);
				}
			}
#pragma unroll
			for (int it = 0; it < NIT; it++) {
				const int item = tid + G::NT * it;
				if (item < ITEMS) {
					const int row = item / IW4, c4 = item - row * IW4;
					const int e[4] = {v[it].x, v[it].y, v[it].z, v[it].w};
					if constexpr (SKIP == 1) {
						T* dst = iiT + row * G::IWp + 4 * c4;
#pragma unroll
						for (int k = 0; k < 4; k++)
							if (4 * c4 + k < G::IW) dst[k] = __builtin_bit_cast(T, e[k]);
					} else {
						// patch column rx = 4 c4 + k - 2: k = 0, 2 are even columns (plane 0), k = 1, 3 odd (plane 1), slots 2 c4 - 1 and 2 c4
						T* p0 = iiT + row * G::IWp + 2 * c4;
						T* p1 = p0 + G::plane;
						if (c4 > 0) { p0[-1] = __builtin_bit_cast(T, e[0]); p1[-1] = __builtin_bit_cast(T, e[1]); }
						if (4 * c4 < G::IW) p0[0] = __builtin_bit_cast(T, e[2]);
						if (4 * c4 + 1 < G::IW) p1[0] = __builtin_bit_cast(T, e[3]);
					}
				}
			}
			staged = true;
		}
	}
	if constexpr (SKIP == 2 && (G::IW % 2 == 0)) {
		// the patch origin is an even column, so with an even row stride every row of the patch is a run of aligned float2 -- half the load
		// instructions of the scalar path below (this kernel is bound by staging: ~9 k floats per tile); the two halves of a pair go to the two
		// column-phase planes at the same index
		if (!staged && !BHIP_ABLATE(P, 4) && (stride & 1) == 0 && (W & 1) == 0 && (((unsigned long long)d) & 7ull) == 0) {
			const int tx = tid & 63, ty = tid >> 6;
			constexpr int PAIRS = G::IW / 2;
			static_assert(PAIRS <= 64, "one lane per float2 of a patch row");
			constexpr int RB = (G::IH + 2 * G::NW - 1) / (2 * G::NW);
			for (int ry0 = ty; ry0 < G::IH; ry0 += G::NW * RB) {
				typedef typename TapVec2<T>::type T2;
				T2 v[RB];
#pragma unroll
				for (int k = 0; k < RB; k++) {
					const int ry = ry0 + G::NW * k;
					const int gy = Y0 + ry;
					const int gx = X0 + 2 * tx;
					const bool ok = ry < G::IH && gy >= 0 && gy < H && tx < PAIRS && gx >= 0 && gx < W;
					const T2* __restrict__ src = (const T2*)(d + (long long)(ok ? gy : 0) * stride + (ok ? gx : 0));
					v[k] = *src;
					if (!ok) { v[k].x = T(0); v[k].y = T(0); }
				}
#pragma unroll
				for (int k = 0; k < RB; k++) {
					const int ry = ry0 + G::NW * k;
					if (ry < G::IH && tx < PAIRS) {
						iiT[ry * G::IWp + tx] = v[k].x;
						iiT[G::plane + ry * G::IWp + tx] = v[k].y;
					}
				}
			}
			staged = true;
		}
	}
	if (!staged && !BHIP_ABLATE(P, 4)) {
		// stage the patch: every thread keeps a batch of independent global loads in flight before the first LDS store
		const int tx = tid & 63, ty = tid >> 6;
		constexpr int COLS = (G::IW + 63) / 64;   // 64-float column chunks per row
		constexpr int RB = (G::IH + 2 * G::NW - 1) / (2 * G::NW);   // rows per batch: the whole patch in two batches of independent loads
		for (int ry0 = ty; ry0 < G::IH; ry0 += G::NW * RB) {
			T v[RB][COLS];
#pragma unroll
			for (int k = 0; k < RB; k++) {
				const int ry = ry0 + G::NW * k;
				const int gy = Y0 + ry;
				const bool rowOk = ry < G::IH && gy >= 0 && gy < H;
				const T* __restrict__ src = d + (long long)(rowOk ? gy : 0) * stride;
#pragma unroll
				for (int cc = 0; cc < COLS; cc++) {
					const int rx = cc * 64 + tx;
					const int gx = X0 + rx;
					const bool ok = rowOk && rx < G::IW && gx >= 0 && gx < W;
					v[k][cc] = src[ok ? gx : 0];
					if (!ok) v[k][cc] = T(0);
				}
			}
#pragma unroll
			for (int k = 0; k < RB; k++) {
				const int ry = ry0 + G::NW * k;
#pragma unroll
				for (int cc = 0; cc < COLS; cc++) {
					const int rx = cc * 64 + tx;
					if (ry < G::IH && rx < G::IW) iiT[G::at(ry, rx)] = v[k][cc];
				}
			}
		}
	}
	__syncthreads();
	// ---- intensity.  The 3x3x3 scale-space test only ever looks at the outer levels (0 and 3) in the 3x3 neighbourhood of a pixel that is
	// already a strict (2R+1)^2 maximum of a mid level (1 or 2), so only the two mid levels are built densely over the tile; the outer
	// levels are evaluated on demand at the few NMS survivors (nine values each).  Same functions of (pixel, kernel size), same decisions.
	static_assert(NL == 4, "two mid levels between two outer levels");
	// the (at most two) mid-level records as wave-uniform values: indexing P.mid[] with a per-lane m is a kernarg load + wait per use
	const FusedMid M0 = P.mid[0], M1 = P.mid[P.nmid > 1 ? 1 : 0];
	float* mid1 = inten;                          // level 1  [ITH][ITp]
	float* mid2 = inten + G::ITH * G::ITp;        // level 2
	int* survList = (int*)(inten + 2 * G::ITH * G::ITp);   // [SURV] packed (m << 16 | py << 8 | px)
	float* sparse = (float*)(survList + G::SURV);          // [SURV][9] outer-level values around each survivor
	int* survCount = (int*)(sparse + G::SPARSE);
	int* candCount = survCount + 1;
	unsigned short* candList = (unsigned short*)sparse;    // [NCAND] pixels above the threshold and their four direct neighbours (m << 12 | py << 6 | px); dead before `sparse` is written
	if (tid == 0) { *survCount = 0; *candCount = 0; }
	if (!BHIP_ABLATE(P, 1)) {
		fusedLevelDense<G, SKIP, R, 1, T>(P, iiT, mid1, tid, x0, y0, X0, Y0);
		fusedLevelDense<G, SKIP, R, 2, T>(P, iiT, mid2, tid, x0, y0, X0, Y0);
	}
	__syncthreads();
	if (P.nexp > 0) {
		// even pixels of the tile core -> pixel (x/2, y/2) of the next octave.  A mid level is read from its tile, an outer level is
		// evaluated here (a quarter of the core's pixels).
		constexpr int HX = (G::TX + 1) / 2, HY = (G::TY + 1) / 2;
		const int ex0 = (x0 + 1) >> 1, ey0 = (y0 + 1) >> 1;   // first even pixel of the core, in next-octave coordinates
		for (int it = tid; it < P.nexp * HX * HY; it += G::NT) {
			const int k = it / (HX * HY);
			const int rem = it - k * (HX * HY);
			const int jy = rem / HX, jx = rem - jy * HX;
			const int ex = ex0 + jx, ey = ey0 + jy;
			const int px = 2 * ex - x0, py = 2 * ey - y0;   // core coordinates in this tile
			if (px < G::TX && py < G::TY && 2 * ex < P.w && 2 * ey < P.h && ex < P.expW && ey < P.expH) {
				const int lv = P.expLevel[k];
				float v;
				if (lv == 1) v = mid1[(py + R) * G::ITp + (px + R)];
				else if (lv == 2) v = mid2[(py + R) * G::ITp + (px + R)];
				else if (lv == 0) v = fusedPixelCall<G, SKIP, 0, T>(P.ii.width, P.ii.height, P.w, P.h, iiT, x0 + px, y0 + py, x0, y0, X0, Y0);
				else v = fusedPixelCall<G, SKIP, 3, T>(P.ii.width, P.ii.height, P.w, P.h, iiT, x0 + px, y0 + py, x0, y0, X0, Y0);
				P.expOut[(long long)img * P.expImageStride + P.expSlotOff[k] + (long long)ey * P.expW + ex] = v;
			}
		}
	}

	// ---- strict (2R+1)^2 maxima of the two mid levels, in two steps so that no wave walks the full window for one or two lanes:
	//   1. thread = one core column of one mid level within a band of rows (a wave = both levels x 32 columns, one band per wave): it
	//      walks down its rows with the value above / at / below in registers -- one LDS read per pixel -- and tests the threshold and
	//      the two vertical neighbours; the few pixels that pass read their two horizontal neighbours.  What is left (a few per cent:
	//      most pixels above the threshold lose against a direct neighbour) is compacted into a workgroup-wide list;
	//   2. one listed pixel per thread: frame tests, the full window, the ignore border -> survivor list.
	{
		static_assert(G::ITW == 32, "a wave = two mid levels x 32 columns");
		constexpr int RB = (G::TY + G::NW - 1) / G::NW;     // rows per band
		const int px = tid & 31, m = (tid >> 5) & 1, band = tid >> 6;
		const int py0 = band * RB;
		if (px < G::TX && m < P.nmid && py0 < G::TY && !BHIP_ABLATE(P, 2)) {
			const float* col = ((m == 0 ? M0.level : M1.level) == 1 ? mid1 : mid2) + (py0 + R) * G::ITp + (px + R);
			float prev = col[-G::ITp], cur = col[0];
#pragma unroll
			for (int k = 0; k < RB; k++) {
				if (py0 + k < G::TY) {
					const float next = col[(k + 1) * G::ITp];
					if (cur >= P.threshold && cur != FLT_MAX && !(prev >= cur) && !(next >= cur)) {
						const float left = col[k * G::ITp - 1], right = col[k * G::ITp + 1];
						// a pixel above its four direct neighbours has none of them in the list: at most every second pixel, NCAND holds them all
						if (!(left >= cur) && !(right >= cur)) candList[atomicAdd(candCount, 1)] = (unsigned short)((m << 12) | ((py0 + k) << 6) | px);
					}
					prev = cur; cur = next;
				}
			}
		}
	}
	__syncthreads();
	{
		const int ncand = *candCount;
		for (int ci = tid; ci < ncand; ci += G::NT) {
			const int cc = candList[ci];
			const int m = cc >> 12, py = (cc >> 6) & 63, px = cc & 63;
			const int code = (m << 16) | (py << 8) | px;
			const float* mid = ((m == 0 ? M0.level : M1.level) == 1 ? mid1 : mid2) + (py + R) * G::ITp + (px + R);
			const float val = mid[0];
			const int b = m == 0 ? M0.border : M1.border;
			const int x = x0 + px, y = y0 + py;
			if (x < b || x >= P.w - b || y < b || y >= P.h - b) continue;
			bool isMax = true;
#pragma unroll
			for (int j = -R; j <= R; j++)
#pragma unroll
				for (int i = -R; i <= R; i++)
					if ((i != 0 || j != 0) && mid[j * G::ITp + i] >= val) isMax = false;
			if (!isMax) continue;
			// candidates hugging the ignore border are dropped (FastHessianFeatureDetector.java:270-276)
			const int ignoreR = b + R;
			if (x < ignoreR || x >= P.w - ignoreR || y < ignoreR || y >= P.h - ignoreR) continue;
			// two strict (2R+1)^2 maxima are more than R apart: at most one per (R+1)^2 block, SURV = 2 * blocks per tile holds them all
			const int slot = atomicAdd(survCount, 1);
			if (slot < G::SURV) survList[slot] = code;
		}
	}
	__syncthreads();
	const int nSurv = min(*survCount, G::SURV);
	for (int base = 0; base < nSurv; base += G::CHUNK) {   // one round for all but the densest tiles
		const int nHere = min(G::CHUNK, nSurv - base);
		// ---- outer level around every survivor: nine values each, one (survivor, neighbour) pair per thread
		for (int it = tid; it < nHere * 9; it += G::NT) {
			const int sv = it / 9, q = it - sv * 9;
			const int code = survList[base + sv];
			const int m = code >> 16, py = (code >> 8) & 0xff, px = code & 0xff;
			const int x = x0 + px + q % 3 - 1, y = y0 + py + q / 3 - 1;
			sparse[it] = (m == 0 ? M0.level : M1.level) == 1 ? fusedPixelCall<G, SKIP, 0, T>(P.ii.width, P.ii.height, P.w, P.h, iiT, x, y, x0, y0, X0, Y0)
															  : fusedPixelCall<G, SKIP, 3, T>(P.ii.width, P.ii.height, P.w, P.h, iiT, x, y, x0, y0, X0, Y0);
		}
		__syncthreads();
		for (int sv = tid; sv < nHere; sv += G::NT) {
			const int code = survList[base + sv];
			const int m = code >> 16, py = (code >> 8) & 0xff, px = code & 0xff;
			const float val = ((m == 0 ? M0.level : M1.level) == 1 ? mid1 : mid2)[(py + R) * G::ITp + (px + R)];
			bool below = true;
#pragma unroll
			for (int q = 0; q < 9; q++)
				if (sparse[sv * 9 + q] >= val) below = false;
			if (m == 0) fusedFinish<G, SKIP, R>(P, M0, mid1, mid2, below, sparse[sv * 9 + 4], px, py, x0 + px, y0 + py, val, img);
			else fusedFinish<G, SKIP, R>(P, M1, mid1, mid2, below, sparse[sv * 9 + 4], px, py, x0 + px, y0 + py, val, img);
		}
		if (base + G::CHUNK < nSurv) __syncthreads();
	}
}

// Picks the tile geometry for one octave; returns false when the ii patch cannot be held in LDS (the caller then runs the unfused kernels).
bool bhip_fused_plan(int skip, int nlevels, const int* sizes, int radius, int* TX, int* TY, int* ldsBytes) {
	int rFmax = 0;
	for (int i = 0; i < nlevels; i++) rFmax = sizes[i] / 2 > rFmax ? sizes[i] / 2 : rFmax;
	const int HR = radius;
	static const int itws[3] = {64, 32, 16};
	static const int tys[3] = {16, 8, 4};
	for (int a = 0; a < 3; a++) {
		const int tx = itws[a] - 2 * HR;
		if (tx < 4) continue;
		for (int bidx = 0; bidx < 3; bidx++) {
			const int ty = tys[bidx];
			const int IW = (tx - 1 + 2 * HR) * skip + 2 * rFmax + 2, IH = (ty - 1 + 2 * HR) * skip + 2 * rFmax + 2;
			const int IWp = IW | 1;
			const int bytes = (IH * IWp + nlevels * (ty + 2 * HR) * (itws[a] + 1)) * 4;
			if (bytes <= 52 * 1024) {  // three workgroups per CU
				*TX = tx; *TY = ty; *ldsBytes = bytes;
				return true;
			}
		}
	}
	return false;
}

// true when the octave runs on the compile-time-geometry kernel (the only fused kernel that can export levels)
bool bhip_fused_is_fixed(int skip, int nlevels, const int* sizes, int radius) {
	if (nlevels != 4 || radius != 2) return false;
#ifdef BHIP_EXPERIMENTS
	{ const char* e = getenv("BHIP_FUSED_ABLATE"); if (e && (atoi(e) & 8)) return false; }
#endif
	const int step = sizes[1] - sizes[0];
	if (sizes[2] - sizes[1] != step || sizes[3] - sizes[2] != step) return false;
	return (skip == 1 && sizes[0] == 9 && step == 6) || (skip == 2 && sizes[0] == 15 && step == 12);
}

int bhip_launch_detect_fused(bhip_ctx* ctx, ImgView ii, int batch, int skip, int nlevels, const int* sizes, int nmid, const DetectLevelParams* mids,
							 const int* midLevels, int radius, float threshold, unsigned int* bitmap, int bitmapWords, KeyPoint* cand, int* candCount,
							 int cap, const FusedExport* exp, bool intTaps) {
	FusedParams P;
	int TX, TY, lds;
	if (!bhip_fused_plan(skip, nlevels, sizes, radius, &TX, &TY, &lds)) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "octave does not fit the fused tile");
	P.ii = ii; P.skip = skip; P.w = ii.width / skip; P.h = ii.height / skip; P.nlevels = nlevels;
	if (P.w <= 0 || P.h <= 0 || nmid <= 0) return BHIP_OK;
	int rFmax = 0;
	for (int i = 0; i < nlevels; i++) {
		const int size = sizes[i];
		FusedLevel& L = P.lv[i];
		L.size = size; L.bS = size / 3; L.bL = size - L.bS - 1; L.rF = size / 2; L.rS = L.bL / 2;
		const int borderOrig = L.rF + 1 + (skip - (L.rF + 1) % skip);
		L.border = borderOrig / skip; L.lost = borderOrig - L.rF - 1;
		L.norm = 1.0f / (float)(size * size);
		const int blockW = size / 3, blockH = size - blockW - 1;
		L.r1 = blockW / 2; L.r2 = blockW + L.r1; L.r3 = blockH / 2; L.b = size / 3;
		rFmax = L.rF > rFmax ? L.rF : rFmax;
	}
	P.TX = TX; P.TY = TY; P.HR = radius; P.ITW = TX + 2 * radius; P.ITH = TY + 2 * radius; P.ITp = P.ITW + 1;
	P.ITWlog = 0;
	while ((1 << P.ITWlog) < P.ITW) P.ITWlog++;
	P.rFmax = rFmax;
	P.IW = (TX - 1 + 2 * radius) * skip + 2 * rFmax + 2;
	P.IH = (TY - 1 + 2 * radius) * skip + 2 * rFmax + 2;
	P.IWp = P.IW | 1;
	P.radius = radius; P.threshold = threshold; P.nmid = nmid; P.batch = batch;
	for (int m = 0; m < nmid; m++) {
		P.mid[m].level = midLevels[m]; P.mid[m].border = mids[m].border; P.mid[m].nbx = mids[m].nbx; P.mid[m].bitBase = mids[m].bitBase;
		P.mid[m].sizeMid = mids[m].sizeMid; P.mid[m].sizeLower = mids[m].sizeLower;
	}
	P.ablate = 0;
#ifdef BHIP_EXPERIMENTS
	{ const char* e = getenv("BHIP_FUSED_ABLATE"); P.ablate = e ? atoi(e) : 0; }
#endif
	P.nexp = 0; P.expOut = nullptr; P.expW = P.expH = 0; P.expImageStride = 0; P.expLevel[0] = P.expLevel[1] = 0; P.expSlotOff[0] = P.expSlotOff[1] = 0;
	if (exp && exp->n > 0) {
		P.nexp = exp->n; P.expLevel[0] = exp->level[0]; P.expLevel[1] = exp->level[1];
		P.expOut = exp->out; P.expW = exp->w; P.expH = exp->h; P.expImageStride = exp->imageStride;
		P.expSlotOff[0] = exp->slotOffset[0]; P.expSlotOff[1] = exp->slotOffset[1];
	}
	P.bitmap = bitmap; P.bitmapWords = bitmapWords; P.cand = cand; P.candCount = candCount; P.cap = cap;
	dim3 grid((P.w + TX - 1) / TX, (P.h + TY - 1) / TY, batch);
	{
		// algorithmic bytes of the fused octave: the integral image read once (nothing else reaches HBM but the key points)
		ProfScope ps(ctx, skip == 1 ? "k_detect_fused_skip1" : "k_detect_fused_skipN", 4.0 * ii.width * ii.height * batch);
		bool launched = false;
		if (nlevels == 4 && radius == 2 && !BHIP_ABLATE(P, 8)) {
			const int step = sizes[1] - sizes[0];
			const bool arith = sizes[2] - sizes[1] == step && sizes[3] - sizes[2] == step;
#ifdef BHIP_EXPERIMENTS
			const char* var = getenv("BHIP_FUSED_VARIANT");   // tile-shape experiments
			const char v = var ? var[0] : 0;
#endif
#define LAUNCH_FIXED(SK, S0, ST, ITWV, TYV, NTV)                                                                                        \
	do {                                                                                                                               \
		typedef FixedGeo<SK, S0, ST, 4, 2, ITWV, TYV, NTV> G;                                                                          \
		const long long nt = (long long)((P.w + G::TX - 1) / G::TX) * ((P.h + G::TY - 1) / G::TY) * batch;                              \
		if (nt > 0x7ffffff0LL) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "batch too large for one fused-octave launch");             \
		dim3 g((unsigned)(((nt + 7) >> 3) << 3));                                                                                      \
		const size_t ldsB = (size_t)G::ldsFloats * 4;                                                                                  \
		if (intTaps) {                                                                                                                 \
			if (ldsB > 65536) BHIP_HIP(ctx, hipFuncSetAttribute((const void*)k_detect_fused_fixed<int, SK, S0, ST, 4, 2, ITWV, TYV, NTV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsB)); \
			hipLaunchKernelGGL((k_detect_fused_fixed<int, SK, S0, ST, 4, 2, ITWV, TYV, NTV>), g, dim3(NTV), ldsB, ctx->stream, P);      \
		} else {                                                                                                                       \
			if (ldsB > 65536) BHIP_HIP(ctx, hipFuncSetAttribute((const void*)k_detect_fused_fixed<float, SK, S0, ST, 4, 2, ITWV, TYV, NTV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsB)); \
			hipLaunchKernelGGL((k_detect_fused_fixed<float, SK, S0, ST, 4, 2, ITWV, TYV, NTV>), g, dim3(NTV), ldsB, ctx->stream, P);    \
		}                                                                                                                              \
		launched = true;                                                                                                               \
	} while (0)
			if (arith && skip == 1 && sizes[0] == 9 && step == 6) {
#ifdef BHIP_EXPERIMENTS   // tile-shape experiments (scripts/variants_fused.sh); the shipped library holds the chosen shapes only
				if (v == 'a') LAUNCH_FIXED(1, 9, 6, 32, 44, 256);
				else if (v == 'x') LAUNCH_FIXED(1, 9, 6, 32, 16, 256);
				else if (v == 'y') LAUNCH_FIXED(1, 9, 6, 32, 60, 512);
				else if (v == 'w') LAUNCH_FIXED(1, 9, 6, 32, 28, 512);
				else if (v == 'z') LAUNCH_FIXED(1, 9, 6, 32, 44, 512);
				else
#endif
				LAUNCH_FIXED(1, 9, 6, 32, 28, 256);   // 28x28 outputs: the (28+4) x 32 intensity tile is exactly 4 passes of 256 threads
			} else if (arith && skip == 2 && sizes[0] == 15 && step == 12) {
#ifdef BHIP_EXPERIMENTS
				if (v == 'a') LAUNCH_FIXED(2, 15, 12, 32, 12, 256);
				else if (v == 'x') LAUNCH_FIXED(2, 15, 12, 32, 20, 256);
				else if (v == 'b') LAUNCH_FIXED(2, 15, 12, 32, 16, 256);   // round-1 shape: 28x16 outputs, four waves
				else if (v == 'w') LAUNCH_FIXED(2, 15, 12, 32, 32, 512);
				else if (v == 'z') LAUNCH_FIXED(2, 15, 12, 32, 24, 512);
				else
#endif
				// 28x40 outputs on eight waves (the tallest tile that leaves two workgroups per CU): 14 staged floats per output instead of 22.9,
				// halo rows 44/40 instead of 20/16, 16 waves per CU instead of 12 -- 3.98 -> 3.1 ms
				LAUNCH_FIXED(2, 15, 12, 32, 40, 512);
			}
#undef LAUNCH_FIXED
		}
		if (!launched) {
			if (intTaps) return bhip_fail(ctx, BHIP_ERR_INVALID, "integer taps need the fixed-geometry fused kernel");
			if (P.nexp > 0) return bhip_fail(ctx, BHIP_ERR_INVALID, "level export needs the fixed-geometry fused kernel");
			hipLaunchKernelGGL(k_detect_fused, grid, dim3(256), (size_t)lds, ctx->stream, P);
		}
	}
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}
