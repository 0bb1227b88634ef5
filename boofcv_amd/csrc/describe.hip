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
// outside the image) and the isInside test disappears.  The orientation angles are ordered by (angle, sample index): the
// reference's ddogleg QuickSort_F64 is unstable and its tie order is not pinned by any reference test (SURVEY hard part 5).
//
// Structure: one wavefront per key point, four per workgroup; every phase keeps many independent loads in flight (the kernel is bound by
// LDS / L2 round trips at 4 waves per SIMD, not by arithmetic):
//   * gathers are branch-free (out-of-bounds samples read a safe address and are zeroed afterwards) and issued in unrolled batches;
//     key points are worked on in coarse-tile order per image and XCD-chunked block order, so the taps of the waves in flight hit L2;
//   * the 289 (angle, index) pairs are sorted by merge path on 32-bit keys with an fp64 check (fp64 merge sort as the fallback);
//   * the sliding-window sweep is enumerated in parallel: c(a) by a coarse fp32 search + exact continuation, the end-pointer schedule by
//     a max-scan, window sums as prefix-sum differences, the (end, owner) pairs by merge path; the reference's serial sweep handles the
//     full-circle regime;
//   * the 81-term sub-region sums keep the reference's sequential fp64 order per output but fetch a whole row of samples and
//     weights per wait.
#include "common.h"
#include <algorithm>
#include <type_traits>

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
	const int* perm;          // optional processing order: slot -> key point index
	// colour SURF (DescribePointSurfPlanar): the descriptor is built band by band from these integral images ([image][band][H][W]),
	// concatenated un-normalised and normalised once; orientation and Laplacian sign still come from `ii` (the grey integral image).
	// nBands == 0: single-band descriptor from `ii`.
	const float* bandData;
	long long bandImageStride, bandStride;
	int nBands;
	double oriRadiusFactor;   // object radius of the orientation = scale * factor (2 = BoofDefaults.SURF_SCALE_TO_RADIUS in the grey
	                          // wrapper, 1 in DetectDescribeSurfPlanar.describe)
	int ldsPerWave;           // bytes
	unsigned long long* stamps; // diagnostic build only: [total][16] cycle stamps
	int sort64;               // BHIP_DESCRIBE_SORT64=1: always sort on the fp64 keys (cross-check of the 32-bit key sort)
	int serialOnly;           // BHIP_DESCRIBE_SERIAL=1: always run the reference's serial window sweep (cross-check of the parallel form)
#ifdef BHIP_EXPERIMENTS
	int stopAfter;            // BHIP_DESCRIBE_STOP=k: waves return at phase boundary k (instruction counts per phase by difference; results are garbage)
	int windowStop;           // BHIP_DESCRIBE_WSTOP=k (8..12): the window sweep returns at its internal boundary k (use with BHIP_DESCRIBE_STOP=3)
#endif
};

#define ORI_EPL_MAX 8     // orientation samples per lane (n <= 512, i.e. radius <= 10)
#define DESC_ROW_MAX 16   // samples per sub-region row (widthSubRegion + 2*overLap)
#define DESC_BATCH 3      // descriptor samples per lane whose 12 taps each are in flight together

// ---- sparse gradient (SparseIntegralGradient_NoBorder_F32) ----
__device__ __forceinline__ int gradRadius(double width) {
	int r = ((int)(width + 0.5)) / 2;
	if (r <= 0) r = 1;
	return r;
}
// Branch-free guarded sample.  The 12 taps of a sample sit at wave-uniform distances from its top-left tap (they depend on the kernel
// radius and the row pitch only), so the wave keeps ten uniform base pointers -- image base + tap distance -- in scalar registers and every
// access of a sample uses the same per-lane byte offset: one address computation per sample instead of one per tap.  A sample whose
// kernel leaves the image (or a lane that is switched off) reads offset 0 -- the top-left tap of the sample at (r+1, r+1), always inside
// when anything is -- and is zeroed afterwards; `anyInside` is false when the image is smaller than the kernel (then every distance is 0:
// only element 0 of the image is ever read, and every sample is zero).
// T = float (GrayF32 integral image: SparseIntegralGradient_NoBorder_F32) or int (GrayS32, the integral image of a GrayU8 frame:
// SparseIntegralGradient_NoBorder_I32.java:46-76 -- integer box differences, handed on as exact values).
// A sample is taken in two steps so that a whole batch of samples has its 10 x batch loads in flight before the first one is waited for:
// gradFetch issues the taps with NO control flow, gradFinish combines them in the reference's order.
template <class T>
struct __attribute__((packed, aligned(4))) Pair2 { T x, y; };   // two neighbouring taps, any 4-byte alignment
template <class T>
struct TapBases {
	// rows y-r-1 (a), y-1 (c1), y (c2), y+r (b); columns x-r-1 (0), x-1 and x (1: fetched as a pair), x+r (3)
	const char *a0, *a1, *a3, *c10, *c13, *c20, *c23, *b0, *b1, *b3;
	unsigned int r1;      // r + 1
	unsigned int spanX;   // W - 2r - 1: a kernel is inside iff (unsigned)(x - r - 1) < spanX && (unsigned)(y - r - 1) < spanY
	unsigned int spanY;
	unsigned int pitch4;  // row pitch in bytes (0 when !anyInside)
	bool anyInside;       // 2r + 2 <= min(W, H)
};
// every lane of the wave holds the same value (one key point per wave): move it to a scalar register, so that what is derived from it
// (tap bases, bounds) stays scalar too
__device__ __forceinline__ unsigned int uniformU32(unsigned int v) { return (unsigned int)__builtin_amdgcn_readfirstlane((int)v); }
// safe8: any 8 readable bytes (the pair loads of an image smaller than the kernel go there; their value is never used)
template <class T>
__device__ __forceinline__ TapBases<T> makeBases(const T* d, int r, int stride, int W, int H, const void* safe8) {
	TapBases<T> B;
	r = (int)uniformU32((unsigned)r);   // (the image base d is uniform already: kernel argument + uniform image index)
	B.anyInside = (2 * r + 2 <= W) && (2 * r + 2 <= H);
	// element distances; all zero when the image is smaller than the kernel (nothing may be read beyond element 0 then)
	const long long st = B.anyInside ? stride : 0, rr = B.anyInside ? r : 0, w = B.anyInside ? 2 * r + 1 : 0;
	const T* a = d;
	const T* c1 = a + rr * st;
	const T* c2 = c1 + st;
	const T* b = c2 + rr * st;
	B.a0 = (const char*)a; B.a1 = B.anyInside ? (const char*)(a + rr) : (const char*)safe8; B.a3 = (const char*)(a + w);
	B.c10 = (const char*)c1; B.c13 = (const char*)(c1 + w);
	B.c20 = (const char*)c2; B.c23 = (const char*)(c2 + w);
	B.b0 = (const char*)b; B.b1 = B.anyInside ? (const char*)(b + rr) : (const char*)safe8; B.b3 = (const char*)(b + w);
	B.r1 = (unsigned)(r + 1);
	B.spanX = B.anyInside ? (unsigned)(W - 2 * r - 1) : 0u;
	B.spanY = B.anyInside ? (unsigned)(H - 2 * r - 1) : 0u;
	B.pitch4 = (unsigned)st * 4u;
	return B;
}
template <class T>
struct GradTaps {
	T p0, p1, p2, p3, p4, p5, p6, p7, p8, p9, p10, p11;
	bool inb;
};
template <class T>
__device__ __forceinline__ T tapAt(const char* base, unsigned int off) { return *(const T*)(base + off); }
template <class T>
__device__ __forceinline__ void gradFetch(const TapBases<T>& B, int x, int y, bool on, GradTaps<T>& t) {
	// isInBounds (SparseScaleGradient.java:48-50): x-r-1 >= 0 && y-r-1 >= 0 && x+r < W && y+r < H, as two unsigned range tests
	const unsigned int ux = (unsigned)x - B.r1, uy = (unsigned)y - B.r1;
	const bool inb = on && ux < B.spanX && uy < B.spanY;
	// 32-bit byte offset of the top-left tap from the image base: an integral image is below 2^32 bytes (checked on the host)
	unsigned int off = inb ? __umul24(uy, B.pitch4) + ux * 4u : 0u;   // uy < 2^15, pitch4 < 2^19 when inb
	// keep the offset a 32-bit register value of its own: the loads below are then "scalar base + zero-extended 32-bit lane offset", which
	// is an addressing mode of global_load; if the compiler is left to fold the select into a 64-bit offset it adds base and offset per tap
	// (buffer loads with the distances as scalar offsets measured 9 % slower than this form, with or without the sc0 policy)
	asm volatile("" : "+v"(off));
	// The accesses are issued row by row (top row: left, centre pair, right; the two middle rows; bottom row): the taps of one row often
	// share a 128-byte line, and with 16 waves of different key points streaming through a CU's 32 KB L1 a line survives a few instructions,
	// not a whole sample.  The two centre columns of the top and of the bottom row are neighbours: one 8-byte request each instead of
	// two 4-byte ones.
	t.p0 = tapAt<T>(B.a0, off);
	const Pair2<T> a = *(const Pair2<T>*)(B.a1 + off);
	t.p1 = a.x; t.p2 = a.y;
	t.p3 = tapAt<T>(B.a3, off);
	__builtin_amdgcn_sched_barrier(0);   // keep the issue order row by row (the loads are independent: the scheduler would group them by width)
	t.p11 = tapAt<T>(B.c10, off); t.p4 = tapAt<T>(B.c13, off);
	__builtin_amdgcn_sched_barrier(0);
	t.p10 = tapAt<T>(B.c20, off); t.p5 = tapAt<T>(B.c23, off);
	__builtin_amdgcn_sched_barrier(0);
	t.p9 = tapAt<T>(B.b0, off);
	const Pair2<T> b = *(const Pair2<T>*)(B.b1 + off);
	t.p8 = b.x; t.p7 = b.y;
	t.p6 = tapAt<T>(B.b3, off);
	t.inb = inb;
}
template <class T>
__device__ __forceinline__ void gradFinish(const GradTaps<T>& t, T& gx, T& gy) {
	const T left = t.p8 - t.p9 - t.p1 + t.p0;
	const T right = t.p6 - t.p7 - t.p3 + t.p2;
	const T top = t.p4 - t.p11 - t.p3 + t.p0;
	const T bottom = t.p6 - t.p9 - t.p5 + t.p10;
	gx = t.inb ? right - left : T(0);
	gy = t.inb ? bottom - top : T(0);
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

// ---------------------------------------------------------------------------------------------------------------
// Stable merge sort of the n (angle, index) pairs in LDS: runs of EPL consecutive elements per lane (insertion sort in registers),
// then log2 rounds in which every element finds its slot by a binary search in the sibling run.  All of a lane's searches advance
// in lock step so each round costs ~log2(L) LDS round trips instead of EPL times that.  On return keyA holds the sorted angles
// and dX, dY are permuted into the same order.
//
// LDS layout of the orientation phase (n samples): while sorting  gX[n] gY[n] (fp32 gradients) | ang[n] (fp64) | 8n bytes of key
// scratch | idxA[n] idxB[n] (u16) = 28n bytes; afterwards the same bytes hold dX[n] dY[n] sA[n] (fp64, sorted) | the window sweep's end-position
// marks (256 * EPLT bytes from 24n on: they overlay the fp32 angle copies / bin offsets, which are dead by then).
// The sorted fp64 samples are rebuilt as (double)g * weight[index] -- the expression the reference evaluates -- so only the
// 4-byte gradients have to live through the sort.
struct OriSortOut {
	double* dX; double* dY; double* sA;
	float* sF;   // (float)sA, for the coarse window searches
};
template <int EPLT, class T>
__device__ __forceinline__ void permuteSorted(const T* gX, const T* gY, const unsigned short* srcI, const double (&a)[EPLT], const double* weights,
											   OriSortOut o, int p0, int cnt) {
	double x[EPLT], y[EPLT];
#pragma unroll
	for (int e = 0; e < EPLT; e++) {
		x[e] = 0.0; y[e] = 0.0;
		if (e < cnt) {
			const int i = srcI[p0 + e];
			x[e] = (double)gX[i]; y[e] = (double)gY[i];
			if (weights) { const double w = weights[i]; x[e] *= w; y[e] *= w; }
		}
	}
	waveSync();
#pragma unroll
	for (int e = 0; e < EPLT; e++)
		if (e < cnt) { o.dX[p0 + e] = x[e]; o.dY[p0 + e] = y[e]; o.sA[p0 + e] = a[e]; o.sF[p0 + e] = (float)a[e]; }
	waveSync();
}

template <int EPLT, class T>
__device__ __forceinline__ void sortSamplesByAngle(const T* gX, const T* gY, double* keyA, double* keyB, unsigned short* idxA, unsigned short* idxB,
													const double* weights, OriSortOut out, int n, int lane) {
	const int EPL = (n + 63) >> 6;   // <= EPLT
	const int p0 = lane * EPL;
	const int cnt = max(0, min(EPL, n - p0));
	{
		double k[EPLT];
		unsigned short id[EPLT];
#pragma unroll
		for (int e = 0; e < EPLT; e++) {
			k[e] = e < cnt ? keyA[p0 + e] : 0.0;
			id[e] = (unsigned short)(p0 + e);
		}
#pragma unroll
		for (int e = 1; e < EPLT; e++) {
#pragma unroll
			for (int f = e; f >= 1; f--) {
				if (f < cnt && k[f - 1] > k[f]) {
					const double tk = k[f]; k[f] = k[f - 1]; k[f - 1] = tk;
					const unsigned short ti = id[f]; id[f] = id[f - 1]; id[f - 1] = ti;
				}
			}
		}
		waveSync();
#pragma unroll
		for (int e = 0; e < EPLT; e++)
			if (e < cnt) { keyA[p0 + e] = k[e]; idxA[p0 + e] = id[e]; }
	}
	waveSync();
	double* srcK = keyA; double* dstK = keyB;
	unsigned short* srcI = idxA; unsigned short* dstI = idxB;
	for (int L = EPL; L < n; L <<= 1) {
		double key[EPLT];
		int lo[EPLT], hi[EPLT], base[EPLT];
		bool right[EPLT];
#pragma unroll
		for (int e = 0; e < EPLT; e++) {
			const int p = p0 + e;
			const int q = p / L;
			const int runStart = q * L;
			right[e] = q & 1;
			const int sibStart = right[e] ? runStart - L : runStart + L;
			lo[e] = min(sibStart, n);
			hi[e] = min(sibStart + L, n);
			key[e] = e < cnt ? srcK[p] : 0.0;
			// destination = pairBase + (p - runStart) + (#sibling elements ordered before key); lo ends as sibStart + that count
			base[e] = (right[e] ? sibStart : runStart) + (p - runStart) - min(sibStart, n);
			if (e >= cnt) hi[e] = lo[e];
		}
		// every step halves [lo,hi): ceil(log2(L+1)) steps settle all searches
		for (int span = L; span > 0; span >>= 1) {
			double v[EPLT];
#pragma unroll
			for (int e = 0; e < EPLT; e++) {
				const int mid = (lo[e] + hi[e]) >> 1;
				v[e] = lo[e] < hi[e] ? srcK[mid] : 0.0;
			}
#pragma unroll
			for (int e = 0; e < EPLT; e++) {
				if (lo[e] < hi[e]) {
					const int mid = (lo[e] + hi[e]) >> 1;
					// left-run elements go before equal right-run elements (stable): strictly-less for left, less-or-equal for right
					const bool before = right[e] ? (v[e] <= key[e]) : (v[e] < key[e]);
					if (before) lo[e] = mid + 1; else hi[e] = mid;
				}
			}
		}
#pragma unroll
		for (int e = 0; e < EPLT; e++) {
			if (e < cnt) {
				const int dst = base[e] + lo[e];
				dstK[dst] = key[e];
				dstI[dst] = srcI[p0 + e];
			}
		}
		waveSync();
		double* tk = srcK; srcK = dstK; dstK = tk;
		unsigned short* ti = srcI; srcI = dstI; dstI = ti;
	}
	{
		double a[EPLT];
#pragma unroll
		for (int e = 0; e < EPLT; e++) a[e] = e < cnt ? srcK[p0 + e] : 0.0;
		permuteSorted<EPLT, T>(gX, gY, srcI, a, weights, out, p0, cnt);
	}
}

// ---------------------------------------------------------------------------------------------------------------
// Fast form of the same sort on 32-bit keys.  (float)angle is a monotone function of the angle, so ordering by the float's bit
// pattern (made unsigned-comparable) with a stable merge can only differ from the fp64 (angle, index) order between elements whose
// floats coincide; those then sit in index order.  After the sort every adjacent pair is checked in fp64: any pair out of order makes
// the function return false and the caller runs the fp64 sort above on the untouched inputs (a fraction of a percent of key points).
// The searches compare integers and move 4-byte keys, roughly half the issue slots of the fp64 version.
__device__ __forceinline__ unsigned int angleKey32(double a) {
	const float f = (float)a + 0.0f;   // -0 -> +0: equal as doubles, must stay tied
	const unsigned int u = __float_as_uint(f);
	return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
template <int EPLT, class T>
__device__ __forceinline__ bool sortSamplesFast32(const T* gX, const T* gY, const double* ang, unsigned int* keyA, unsigned int* keyB,
												   unsigned short* idxA, unsigned short* idxB, const double* weights, OriSortOut out, int n, int lane) {
	const int EPL = (n + 63) >> 6;   // <= EPLT
	const int p0 = lane * EPL;
	const int cnt = max(0, min(EPL, n - p0));
	{
		unsigned int k[EPLT];
		unsigned short id[EPLT];
#pragma unroll
		for (int e = 0; e < EPLT; e++) {
			k[e] = e < cnt ? angleKey32(ang[p0 + e]) : 0xffffffffu;
			id[e] = (unsigned short)(p0 + e);
		}
#pragma unroll
		for (int e = 1; e < EPLT; e++) {
#pragma unroll
			for (int f = e; f >= 1; f--) {
				if (f < cnt && k[f - 1] > k[f]) {
					const unsigned int tk = k[f]; k[f] = k[f - 1]; k[f - 1] = tk;
					const unsigned short ti = id[f]; id[f] = id[f - 1]; id[f - 1] = ti;
				}
			}
		}
#pragma unroll
		for (int e = 0; e < EPLT; e++)
			if (e < cnt) { keyA[p0 + e] = k[e]; idxA[p0 + e] = id[e]; }
	}
	waveSync();
	unsigned int* srcK = keyA; unsigned int* dstK = keyB;
	unsigned short* srcI = idxA; unsigned short* dstI = idxB;
	// Merge rounds by merge path: a lane owns EPL consecutive OUTPUT slots of its pair of runs, finds where its diagonal cuts the two
	// runs with ONE binary search, and merges its EPL outputs serially (A before B on ties = stable).  About a third of the
	// instructions of one binary search per element.
	for (int L = EPL; L < n; L <<= 1) {
		const int s = (p0 / (2 * L)) * (2 * L);
		const int a0 = s, b0 = s + L;
		const int lenA = cnt > 0 ? min(L, n - s) : 0;
		const int lenB = cnt > 0 ? max(0, min(L, n - b0)) : 0;
		const int d = p0 - s;
		int lo = max(0, d - lenB), hi = min(d, lenA);
		for (int span = L; span > 0; span >>= 1) {
			const int mid = (lo + hi) >> 1;
			const bool on = lo < hi;
			const unsigned int ka = on ? srcK[a0 + mid] : 0u;
			const unsigned int kb = on ? srcK[b0 + d - 1 - mid] : 0u;
			if (on) { if (ka <= kb) lo = mid + 1; else hi = mid; }
		}
		int i = lo, j = d - lo;
		unsigned int ka = i < lenA ? srcK[a0 + i] : 0xffffffffu;
		unsigned int kb = j < lenB ? srcK[b0 + j] : 0xffffffffu;
		unsigned int outK[EPLT];
		int sp[EPLT];
#pragma unroll
		for (int e = 0; e < EPLT; e++) {
			const bool takeA = ka <= kb;   // an exhausted run shows the sentinel; real keys are below it
			outK[e] = takeA ? ka : kb;
			sp[e] = takeA ? a0 + i : b0 + j;
			i += takeA ? 1 : 0;
			j += takeA ? 0 : 1;
			const int nxt = takeA ? a0 + i : b0 + j;
			const int lim = takeA ? a0 + lenA : b0 + lenB;
			const unsigned int nk = (e + 1 < EPLT && nxt < lim) ? srcK[nxt] : 0xffffffffu;
			if (takeA) ka = nk; else kb = nk;
		}
		unsigned short outI[EPLT];
#pragma unroll
		for (int e = 0; e < EPLT; e++) outI[e] = e < cnt ? srcI[sp[e]] : (unsigned short)0;
#pragma unroll
		for (int e = 0; e < EPLT; e++)
			if (e < cnt) { dstK[p0 + e] = outK[e]; dstI[p0 + e] = outI[e]; }
		waveSync();
		unsigned int* tk = srcK; srcK = dstK; dstK = tk;
		unsigned short* ti = srcI; srcI = dstI; dstI = ti;
	}
	// fp64 check of every adjacent pair, then the permutation of the samples into sorted order
	double a[EPLT];
	bool bad = false;
	{
		int id[EPLT], idn[EPLT];
		double an[EPLT];
#pragma unroll
		for (int e = 0; e < EPLT; e++) {
			id[e] = e < cnt ? srcI[p0 + e] : 0;
			idn[e] = (e < cnt && p0 + e + 1 < n) ? srcI[p0 + e + 1] : -1;
		}
#pragma unroll
		for (int e = 0; e < EPLT; e++) {
			a[e] = ang[id[e]];
			an[e] = idn[e] >= 0 ? ang[idn[e]] : 0.0;
		}
#pragma unroll
		for (int e = 0; e < EPLT; e++)
			if (idn[e] >= 0 && (a[e] > an[e] || (a[e] == an[e] && id[e] > idn[e]))) bad = true;
	}
	if (__any(bad)) return false;
	permuteSorted<EPLT, T>(gX, gY, srcI, a, weights, out, p0, cnt);
	return true;
}

// ---------------------------------------------------------------------------------------------------------------
// Counting sort of the (angle, index) pairs -- the normal path.  bin(angle) is a monotone function of the angle (float conversion, add,
// multiply by a positive constant, truncation: each monotone), so ordering by bin and then exactly by (fp64 angle, index) inside a bin IS
// the (angle, index) order of the merge sorts above.  ORI_BINS bins of 2*pi/1024 rad hold 0.3 samples on average: a lane counts its
// samples into packed 16-bit LDS counters, one wave-wide exclusive scan turns the counts into bin offsets, every sample ranks itself
// against the other members of its bin, and is scattered straight from registers to its sorted slot (the samples never visit LDS
// unsorted).  The offsets stay in LDS: the window search below reads "how many samples lie below angle x" from them instead of
// binary-searching the sorted angles.  A bin with more than ORI_BIN_MAX members (a run of equal angles: samples outside the image have a
// zero gradient, angle 0) makes the function return false; the caller then takes the merge-sort path.
#define ORI_BINS 1024
#define ORI_BIN_MAX 8
__device__ __forceinline__ int angleBin(double a) {
	const float f = ((float)a + 3.14159274f) * ((float)ORI_BINS / 6.28318548f);
	return min(max((int)f, 0), ORI_BINS - 1);
}
__device__ __forceinline__ int binOffset(const unsigned int* bins, int b) { return (int)((bins[b >> 1] >> ((b & 1) * 16)) & 0xffffu); }

template <int EPLT>
__device__ __forceinline__ bool countingSortScatter(const double (&a)[EPLT], const double (&dx)[EPLT], const double (&dy)[EPLT], int n, int lane,
													 unsigned int* bins, double* tmpA, unsigned short* tmpI, double* dX, double* dY, double* sA) {
	static_assert(ORI_BINS == 1024, "two 16-byte stores per lane clear the counters");
	{
		const uint4 z = make_uint4(0, 0, 0, 0);
		reinterpret_cast<uint4*>(bins)[2 * lane] = z;
		reinterpret_cast<uint4*>(bins)[2 * lane + 1] = z;
	}
	waveSync();
	int bin[EPLT], r[EPLT];
#pragma unroll
	for (int e = 0; e < EPLT; e++) {
		bin[e] = angleBin(a[e]);
		r[e] = 0;
		if (lane + 64 * e < n) {
			const int sh = (bin[e] & 1) * 16;
			const unsigned int old = atomicAdd(&bins[bin[e] >> 1], 1u << sh);
			r[e] = (int)((old >> sh) & 0xffffu);
		}
	}
	waveSync();
	int cnt[EPLT];
	int mx = 0;
#pragma unroll
	for (int e = 0; e < EPLT; e++) {
		cnt[e] = lane + 64 * e < n ? binOffset(bins, bin[e]) : 0;
		mx = max(mx, cnt[e]);
	}
#pragma unroll
	for (int o = 32; o >= 1; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
	if (mx > ORI_BIN_MAX) return false;   // wave-uniform
	waveSync();   // every count has been read before the counters turn into offsets
	{
		// exclusive scan over the 1024 counters: 16 per lane (8 words), then across the lanes
		const uint4 w0 = reinterpret_cast<const uint4*>(bins)[2 * lane], w1 = reinterpret_cast<const uint4*>(bins)[2 * lane + 1];
		const unsigned int w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
		unsigned int pre[16];
		unsigned int run = 0;
#pragma unroll
		for (int i = 0; i < 8; i++) {
			pre[2 * i] = run; run += w[i] & 0xffffu;
			pre[2 * i + 1] = run; run += w[i] >> 16;
		}
		unsigned int sc = run;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const unsigned int t = __shfl_up(sc, o, 64);
			if (lane >= o) sc += t;
		}
		const unsigned int base = sc - run;
		unsigned int ow[8];
#pragma unroll
		for (int i = 0; i < 8; i++) ow[i] = (base + pre[2 * i]) | ((base + pre[2 * i + 1]) << 16);
		reinterpret_cast<uint4*>(bins)[2 * lane] = make_uint4(ow[0], ow[1], ow[2], ow[3]);
		reinterpret_cast<uint4*>(bins)[2 * lane + 1] = make_uint4(ow[4], ow[5], ow[6], ow[7]);
	}
	waveSync();
	int off[EPLT];
#pragma unroll
	for (int e = 0; e < EPLT; e++) {
		off[e] = 0;
		if (lane + 64 * e < n) {
			off[e] = binOffset(bins, bin[e]);
			tmpA[off[e] + r[e]] = a[e];
			tmpI[off[e] + r[e]] = (unsigned short)(lane + 64 * e);
		}
	}
	waveSync();
	int rank[EPLT];
#pragma unroll
	for (int e = 0; e < EPLT; e++) rank[e] = 0;
	if (mx > 1) {
		for (int k = 0; k < mx; k++) {
			double ak[EPLT];
			int ik[EPLT];
#pragma unroll
			for (int e = 0; e < EPLT; e++) {
				const bool on = k < cnt[e];
				ak[e] = on ? tmpA[off[e] + k] : 0.0;
				ik[e] = on ? (int)tmpI[off[e] + k] : 0;
			}
#pragma unroll
			for (int e = 0; e < EPLT; e++)
				if (k < cnt[e] && (ak[e] < a[e] || (ak[e] == a[e] && ik[e] < lane + 64 * e))) rank[e]++;
		}
	}
	waveSync();   // tmpA / tmpI are dead: the sorted arrays may overwrite them
#pragma unroll
	for (int e = 0; e < EPLT; e++)
		if (lane + 64 * e < n) {
			const int f = off[e] + rank[e];
			dX[f] = dx[e]; dY[f] = dy[e]; sA[f] = a[e];
		}
	waveSync();
	return true;
}

// ---------------------------------------------------------------------------------------------------------------
// Sliding-window orientation, wave-parallel form of ImplOrientationSlidingWindowIntegral.estimateAngle (:139-188).
//
// The reference sorts the n angles and sweeps two pointers: for every start a the window grows while
// UtilAngle.dist(angle[a], angle[end]) <= windowSize, keeping running sums that gain the element at `end` and later lose the
// element at `start`; after every gain it tests |sum|^2 > best.  Which elements are in the window at each test depends only on
// the sorted angles, so the candidate windows are enumerated in parallel:
//   c(a)  = number of consecutive successors of a inside its window (binary search: the predicate is monotone on the leading side)
//   E(a)  = max(E(a-1), a + c(a) + 1), E(-1) = 1          position of `end` after start a (a max-scan over the wave)
//   the test made right after element E joins belongs to start a(E) = min{a : E(a) > E} and sees the sorted elements a(E)..E
// Every test the reference makes is made here on the same set of elements; the window sum is taken as a difference of prefix sums
// over the sorted samples instead of the reference's running add/subtract chain, so the two differ only by fp64 rounding of the
// sums (~1e-15 relative), far inside the 1e-5 descriptor bar.  The first maximum in sweep order wins, as in the reference.
// The full-circle regime (some window wraps all the way round: ramps, flat patches) is detected and left to the serial code.
// Returns false when the caller must run the serial sweep (the arrays are then still the sorted samples).
template <int EPLT>
__device__ __forceinline__ bool slidingWindowFast(double* dX, double* dY, const double* sA, const float* sF, const unsigned int* bins, unsigned int* marks, int n,
												   double window, int lane, double& bestX, double& bestY, unsigned long long* st /*diagnostic stamps or nullptr*/, int wstop = -1) {
#ifdef BHIP_EXPERIMENTS
#define WSTAMP(i) do { if (st && lane == 0) st[i] = __builtin_readcyclecounter(); if (wstop == (i)) { bestX = 1.0; bestY = 0.0; return true; } } while (0)
#else
#define WSTAMP(i) do { if (st && lane == 0) st[i] = __builtin_readcyclecounter(); } while (0)
#endif
	const int EPL = (n + 63) >> 6;   // <= EPLT
	const int p0 = lane * EPL;
	const int cnt = max(0, min(EPL, n - p0));
	bool abnormal = false;
	int valE[EPLT];   // a + c(a) + 1
	int runMax = 0;
	{
		// c(a): largest t such that every successor 1..t is on the leading side and inside the window.  Coarse part: binary search on
		// the fp32 copies for the largest t that is inside by a margin (forward offset <= window - 4e-6; the fp32 offset is within
		// 1.4e-6 of the exact one, so everything up to there passes the exact test as well).  Exact part: the reference's own test
		// (UtilAngle.dist on the fp64 angles) walks on from there -- normally it fails at once.
		const float winLo = (float)window - 4.0e-6f;
		const float twoPiF = 6.28318530717958647692f;
		if (!(winLo > 0.0f)) return false;
		float fa[EPLT];
		int lo[EPLT], hi[EPLT];
		if (bins) {
			// coarse part from the counting sort's offsets: every sample in a bin below bin(T), T = angle + window - 1e-9, has an angle below
			// T (the bin function is monotone), i.e. is inside the window by a margin far above fp64 rounding.  offsets[bin(T)] is the number
			// of such samples; past +pi the window continues from -pi.
#pragma unroll
			for (int e = 0; e < EPLT; e++) {
				lo[e] = 0;
				if (e < cnt) {
					const int a = p0 + e;
					const double T = sA[a] + (window - 1.0e-9);
					const bool wrap = T >= M_PI;
					const int below = binOffset(bins, angleBin(wrap ? T - 2.0 * M_PI : T));
					const int c0 = (wrap ? n + below : below) - (a + 1);
					lo[e] = min(max(c0, 0), n - 1);
				}
			}
		} else {
#pragma unroll
		for (int e = 0; e < EPLT; e++) {
			fa[e] = e < cnt ? sF[p0 + e] : 0.0f;
			lo[e] = 0;
			hi[e] = e < cnt ? n : 1;
		}
		for (int span = n; span > 1; span = (span + 1) >> 1) {
			float fk[EPLT];
#pragma unroll
			for (int e = 0; e < EPLT; e++) {
				const int mid = (lo[e] + hi[e]) >> 1;
				const int kabs = p0 + e + mid;
				const int k = kabs >= n ? kabs - n : kabs;
				fk[e] = (hi[e] - lo[e] > 1) ? sF[k] : 0.0f;
			}
#pragma unroll
			for (int e = 0; e < EPLT; e++) {
				if (hi[e] - lo[e] > 1) {
					const int mid = (lo[e] + hi[e]) >> 1;
					const int kabs = p0 + e + mid;
					const float fo = (fk[e] - fa[e]) + (kabs >= n ? twoPiF : 0.0f);
					if (fo <= winLo) lo[e] = mid; else hi[e] = mid;
				}
			}
		}
		}
		double ta[EPLT], nxt[EPLT];
		int cc[EPLT];
		bool act[EPLT];
#pragma unroll
		for (int e = 0; e < EPLT; e++) {
			ta[e] = e < cnt ? sA[p0 + e] : 0.0;
			cc[e] = lo[e];
			act[e] = e < cnt && lo[e] < n - 1;
			const int kabs = p0 + e + lo[e] + 1;
			const int k = kabs >= n ? kabs - n : kabs;
			nxt[e] = act[e] ? sA[k] : 0.0;
			if (e < cnt && !act[e]) abnormal = true;   // the coarse count alone already spans the whole list: full-circle regime
		}
		// exact continuation, all of a lane's starts in lock step (their LDS reads overlap): the reference's own test walks on from the
		// coarse count -- normally it fails at once; a handful of steps at most unless many samples sit within 1e-9 of the window edge
		// One step = the reference's test `UtilAngle.dist(angle[start], angle[end]) <= windowSize` on the next successor, kept to the
		// leading side.  With d1 = fl(next - start) (= -fl(start - next) exactly) and fo = d1, or fl(d1 + 2 pi) for a successor index that
		// has wrapped past the end of the sorted list, UtilAngle.dist evaluates to fo whenever fo < pi, and the window is below pi (the
		// caller checks < 3), so "leading side and inside" is exactly fo <= window.  A successor that fails it but lies within the window
		// BEHIND the start -- dist = fl(2 pi - d1) unwrapped, -d1 wrapped -- would keep the reference's sweep going round the circle
		// (UtilAngle.dist is symmetric): not a leading-side window, left to the serial sweep.
		for (int guard = 0; guard < 18; guard++) {
			bool any = false;
#pragma unroll
			for (int e = 0; e < EPLT; e++) {
				if (act[e]) {
					const bool wrapped = p0 + e + cc[e] + 1 >= n;
					const double d1 = nxt[e] - ta[e];
					const double fo = wrapped ? d1 + 2.0 * M_PI : d1;
					if (!(fo <= window)) {
						const double behind = wrapped ? -d1 : fabs(2.0 * M_PI + (-d1));
						if (behind <= window) abnormal = true;
						act[e] = false;
					} else {
						cc[e]++;
						if (cc[e] >= n - 1) { abnormal = true; act[e] = false; }
						else if (guard == 17) { abnormal = true; act[e] = false; }   // a long run of samples at the window edge
					}
				}
			}
#pragma unroll
			for (int e = 0; e < EPLT; e++) {
				if (act[e]) {
					const int k2abs = p0 + e + cc[e] + 1;
					nxt[e] = sA[k2abs >= n ? k2abs - n : k2abs];
					any = true;
				}
			}
			if (!__any(any)) break;
		}
#pragma unroll
		for (int e = 0; e < EPLT; e++) {
			valE[e] = 0;
			if (e < cnt) {
				valE[e] = p0 + e + cc[e] + 1;
				runMax = max(runMax, valE[e]);
			}
		}
	}
	WSTAMP(8);
	// inclusive max-scan of the per-lane maxima, then the exclusive value for this lane (E(-1) = 1)
	int scan = runMax;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const int t = __shfl_up(scan, o, 64);
		if (lane >= o) scan = max(scan, t);
	}
	int prevE = __shfl_up(scan, 1, 64);
	if (lane == 0) prevE = 1;
	prevE = max(prevE, 1);
	// ---- validate the schedule (full-circle regime -> serial code), before the samples are overwritten by their prefix sums
	int lastE = 1;
	int Ereg[EPLT];   // E(a) of this lane's starts
	{
		int pe = prevE;
		double chk[EPLT];
		bool need[EPLT];
#pragma unroll
		for (int e = 0; e < EPLT; e++) {
			need[e] = false;
			chk[e] = 0.0;
			Ereg[e] = 0;
			if (e < cnt) {
				const int a = p0 + e;
				const int Ea = max(pe, valE[e]);
				if (pe > valE[e] - 1 && pe > a && pe < a + n) {
					// `end` is already beyond this start's leading window: the reference tests it once and it must fail
					const int k = pe >= n ? pe - n : pe;
					need[e] = true;
					chk[e] = sA[k];
				}
				if (Ea >= a + n) abnormal = true;
				pe = Ea;
				Ereg[e] = Ea;
			}
		}
#pragma unroll
		for (int e = 0; e < EPLT; e++)
			if (need[e] && angleDist(sA[p0 + e], chk[e]) <= window) abnormal = true;
		lastE = pe;
	}
	if (__any(abnormal)) return false;
	lastE = __shfl(lastE, (n - 1) / EPL, 64);  // E(n-1): one past the last end position that is ever tested
	WSTAMP(9);
	// ---- inclusive prefix sums of (dX, dY) in sorted order, in place: window sum(a..E) = I[E] - I[a-1]
	{
		double lx[EPLT], ly[EPLT], vx[EPLT], vy[EPLT];
#pragma unroll
		for (int e = 0; e < EPLT; e++) {
			vx[e] = e < cnt ? dX[p0 + e] : 0.0;
			vy[e] = e < cnt ? dY[p0 + e] : 0.0;
		}
		double tx = 0, ty = 0;
#pragma unroll
		for (int e = 0; e < EPLT; e++) {
			tx += vx[e]; ty += vy[e];
			lx[e] = tx; ly[e] = ty;
		}
		double ox = tx, oy = ty;   // inclusive scan of the lane totals
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const double ux = __shfl_up(ox, o, 64), uy = __shfl_up(oy, o, 64);
			if (lane >= o) { ox += ux; oy += uy; }
		}
		ox -= tx; oy -= ty;        // exclusive
		waveSync();
#pragma unroll
		for (int e = 0; e < EPLT; e++)
			if (e < cnt) { dX[p0 + e] = ox + lx[e]; dY[p0 + e] = oy + ly[e]; }
	}
	waveSync();
	const double totX = dX[n - 1], totY = dY[n - 1];
	WSTAMP(10);
	// ---- candidate windows, one per end position E in [1, E(n-1))
	double bMag = -1.0, bX = 0, bY = 0;
	int bPos = 0x7fffffff;
	if (lane == 0) { bX = dX[0]; bY = dY[0]; bMag = bX * bX + bY * bY; bPos = 0; }
	// Owner of end position E: a(E) = min{a : E(a) > E} = #{a <= n-2 : E(a) <= E} (E(a) never decreases; E(a) > a always, so the owner
	// is at most n-1).  Every start drops a mark at its E(a) (packed 16-bit counters, as in the counting sort), an inclusive scan over the
	// end positions turns the marks into owners, and the candidate windows -- one per end position in [1, E(n-1)) -- are dealt out 64 at a
	// time, each with four independent reads of the prefix sums.  `marks` has room for 2 * 64 * EPLT >= 2n end positions.
	{
#pragma unroll
		for (int e = 0; e < EPLT; e++) marks[lane * EPLT + e] = 0u;
		waveSync();
#pragma unroll
		for (int e = 0; e < EPLT; e++)
			if (e < cnt && p0 + e <= n - 2) atomicAdd(&marks[Ereg[e] >> 1], 1u << ((Ereg[e] & 1) * 16));
		waveSync();
		{
			unsigned int w[EPLT], pre[2 * EPLT];
			unsigned int run = 0;
#pragma unroll
			for (int e = 0; e < EPLT; e++) w[e] = marks[lane * EPLT + e];
#pragma unroll
			for (int e = 0; e < EPLT; e++) {
				run += w[e] & 0xffffu; pre[2 * e] = run;
				run += w[e] >> 16; pre[2 * e + 1] = run;
			}
			unsigned int sc = run;
#pragma unroll
			for (int o = 1; o < 64; o <<= 1) {
				const unsigned int t = __shfl_up(sc, o, 64);
				if (lane >= o) sc += t;
			}
			const unsigned int base = sc - run;
#pragma unroll
			for (int e = 0; e < EPLT; e++) marks[lane * EPLT + e] = (base + pre[2 * e]) | ((base + pre[2 * e + 1]) << 16);
		}
		waveSync();
		const unsigned short* owner = (const unsigned short*)marks;
		for (int E0 = 1; E0 < lastE; E0 += 64) {
			const int E = E0 + lane;
			if (E < lastE) {
				const int a = owner[E];
				const int ke = E >= n ? E - n : E;
				const double ax = a > 0 ? dX[a - 1] : 0.0, ay = a > 0 ? dY[a - 1] : 0.0;
				const double ex = dX[ke], ey = dY[ke];
				double sx, sy;
				if (E < n) { sx = ex - ax; sy = ey - ay; }
				else { sx = (totX - ax) + ex; sy = (totY - ay) + ey; }
				const double mag = sx * sx + sy * sy;
				if (mag > bMag) { bMag = mag; bX = sx; bY = sy; bPos = E; }   // E increases within a lane: strict > keeps the first
			}
		}
	}
	WSTAMP(11);
	// ---- first maximum in sweep order across the wave: max magnitude, then the smallest position among the lanes that hold it
	double m = bMag;
#pragma unroll
	for (int o = 32; o >= 1; o >>= 1) m = fmax(m, __shfl_xor(m, o, 64));
	int pos = (bMag == m) ? bPos : 0x7fffffff;
#pragma unroll
	for (int o = 32; o >= 1; o >>= 1) pos = min(pos, __shfl_xor(pos, o, 64));
	const unsigned long long owner = __ballot(bMag == m && bPos == pos);
	const int src = __ffsll((long long)owner) - 1;
	bestX = __shfl(bX, src, 64);
	bestY = __shfl(bY, src, 64);
	WSTAMP(12);
	return true;
#undef WSTAMP
}

// STAMP = true is a diagnostic build (BHIP_DESCRIBE_STAMPS): lane 0 of every wave records the cycle counter at the phase boundaries
// into a buffer nothing else reads; never used for results or for quoted run times.
#ifdef BHIP_EXPERIMENTS
#define DSTAMP(i) do { if (STAMP && lane == 0) P.stamps[g * 16 + (i)] = __builtin_readcyclecounter(); if (P.stopAfter == (i)) return; } while (0)
#else
#define DSTAMP(i) do { if (STAMP && lane == 0) P.stamps[g * 16 + (i)] = __builtin_readcyclecounter(); } while (0)
#endif

// EPLT = orientation samples per lane = ceil(n / 64), TWT = samples per sub-region row: compile-time for the common configurations so
// the unrolled batches carry no dead slots (the kernel is issue bound); <8,16> is the generic instantiation.
// CFG: 0 = every size comes from the tables at run time; 1 = FactoryDetectDescribe.surfStable defaults (4x4 grid of 5-sample sub-regions,
// overlap 2, 17x17 sliding-window orientation grid); 2 = surfFast defaults (4x4 grid of 5, no overlap, 13x13 average orientation).  With
// the sizes known the index arithmetic (sample -> row/column, feature -> sub-region) folds to shifts and multiplies.
template <bool STAMP, int EPLT, int TWT, class TAP, int CFG = 0>
__global__ __launch_bounds__(256) void k_describe(DescParams P) {
	extern __shared__ __attribute__((aligned(16))) unsigned char ldsAll[];
	// the wave index is uniform by construction; telling the compiler so keeps everything derived from the key point (image, scale, kernel
	// radius, tap strides) in scalar registers and the gathers in base + 32-bit-offset form
	const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int lane = threadIdx.x & 63;
	// XCD-aware order (blocks are dealt round-robin over the 8 XCDs): block b takes key-point group (b % 8) * chunk + b / 8, so each
	// XCD works through its own contiguous run of key points -- one or two images at a time in its L2 instead of the whole batch
	const long long nblk = (P.total + 3) >> 2;
	const long long chunk = (nblk + 7) >> 3;
	const long long blk = (long long)(blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
	if ((long long)(blockIdx.x >> 3) >= chunk) return;
	const long long slot = blk * 4 + wave;
	if (slot >= P.total) return;
	const long long g = P.perm ? (long long)P.perm[slot] : slot;   // processing order (detect.hip, k_kp_tile_*); results stay at index g
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
	img = (int)uniformU32((unsigned)img);
	local = (int)uniformU32((unsigned)local);
	const KeyPoint kp = P.imageStart ? P.kps[(long long)img * P.cap + local] : P.kps[local];
	const TAP* __restrict__ d = (const TAP*)P.ii.data + (long long)img * P.ii.imageStride;
	const int stride = P.ii.stride, W = P.ii.width, H = P.ii.height;
	const SurfTables& T = P.t;

	DSTAMP(0);
	// ------------------------------------------------------------------ orientation
	double angle;
	double dirX = 0.0, dirY = 0.0;   // the vector whose atan2 is the orientation (when it is computed here)
	if (P.anglesIn) {
		angle = P.anglesIn[g];
	} else {
		const double radius = kp.scale * P.oriRadiusFactor;
		const double oscale = radius * T.oriRadiusToScale;    // setObjectRadius
		const TapBases<TAP> G = makeBases<TAP>(d, gradRadius(oscale * T.oriKernelWidth), stride, W, H, P.kps);
		const double period = oscale * T.oriPeriod;
		double tl_x = kp.x - T.oriRadius * period;
		double tl_y = kp.y - T.oriRadius * period;
		tl_x += 0.5;
		tl_y += 0.5;
		const bool oriSliding = CFG == 1 ? true : CFG == 2 ? false : (T.oriStable != 0);
		const int sw = CFG == 1 ? 17 : CFG == 2 ? 13 : T.oriWidth, n = sw * sw;
		// sort-phase layout (see sortSamplesByAngle); the average variant only stages its addends as dX, dY
		TAP* gX = (TAP*)lds;
		TAP* gY = gX + n;
		double* ang = (double*)(lds + (size_t)8 * n);
		double* keyB = (double*)(lds + (size_t)16 * n);
		double* dX = (double*)lds;
		double* dY = dX + n;
		// all of this lane's samples: taps first (independent loads in flight together), then the fp64 tail.  The sliding-window variant
		// keeps its samples (weighted gradient, angle) in registers: the counting sort scatters them straight to their sorted slots.
		TAP gx[EPLT], gy[EPLT];
		double dxr[EPLT], dyr[EPLT], ar[EPLT];
		{
			GradTaps<TAP> tp[EPLT];
#pragma unroll
			for (int e = 0; e < EPLT; e++) {
				const int idx = lane + 64 * e;
				const int sy = idx / sw, sx = idx - sy * sw;
				const int xx = (int)(tl_x + sx * period);
				const int yy = (int)(tl_y + sy * period);
				gradFetch<TAP>(G, xx, yy, idx < n, tp[e]);
			}
#pragma unroll
			for (int e = 0; e < EPLT; e++) gradFinish<TAP>(tp[e], gx[e], gy[e]);
		}
#pragma unroll
		for (int e = 0; e < EPLT; e++) {
			const int idx = lane + 64 * e;
			dxr[e] = 0.0; dyr[e] = 0.0; ar[e] = 0.0;
			if (idx < n) {
				double dx = (double)gx[e], dy = (double)gy[e];
				if (oriSliding) {
					if (T.oriHasWeights) {
						const double w = T.oriWeights[idx];
						dx *= w;
						dy *= w;
					}
					dxr[e] = dx; dyr[e] = dy;
					ar[e] = atan2(dy, dx);
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
		}
		waveSync();
		DSTAMP(1);
		if (oriSliding) {
			unsigned short* idxA = (unsigned short*)(lds + (size_t)24 * n);
			unsigned short* idxB = idxA + n;
			double bestX = 0, bestY = 0;
			bool needSerial = true;
			// n <= 64 * EPLT is enforced on the host
			const OriSortOut so{dX, dY, dY + n, (float*)(lds + (size_t)28 * n)};
			const double* wts = T.oriHasWeights ? T.oriWeights : nullptr;
			unsigned int* bins = (unsigned int*)(lds + (((size_t)28 * n + 15) & ~(size_t)15));
			const bool counted = !P.sort64 && countingSortScatter<EPLT>(ar, dxr, dyr, n, lane, bins, (double*)lds, (unsigned short*)(lds + (size_t)8 * n), dX, dY, dY + n);
			if (!counted) {
				// merge-sort path (a crowded bin, or the BHIP_DESCRIBE_SORT64 cross-check): the unsorted samples go to LDS first
#pragma unroll
				for (int e = 0; e < EPLT; e++) {
					const int idx = lane + 64 * e;
					if (idx < n) { gX[idx] = gx[e]; gY[idx] = gy[e]; ang[idx] = ar[e]; }
				}
				waveSync();
				bool sorted = false;
				if (!P.sort64) sorted = sortSamplesFast32<EPLT, TAP>(gX, gY, ang, (unsigned int*)keyB, (unsigned int*)keyB + n, idxA, idxB, wts, so, n, lane);
				if (!sorted) sortSamplesByAngle<EPLT, TAP>(gX, gY, ang, keyB, idxA, idxB, wts, so, n, lane);
			}
			ang = so.sA;   // sorted angles; dX, dY hold the sorted samples
			DSTAMP(2);
			if (T.oriWindow < 3.0 && !P.serialOnly) needSerial = !slidingWindowFast<EPLT>(dX, dY, ang, so.sF, counted ? bins : nullptr, (unsigned int*)(lds + (size_t)24 * n), n, T.oriWindow, lane, bestX, bestY, STAMP ? P.stamps + g * 16 : nullptr
#ifdef BHIP_EXPERIMENTS
																				  , P.windowStop
#endif
																				  );
			if (needSerial) {
				// estimateAngle() exactly as written in the reference, on the arrays already in sorted order (order[k] == k).
				// Reached for the full-circle regime (all gradients within one window of each other: ramps, flat patches).
				if (lane == 0) {
					const int total = n;
					int start = 0, end = 1;
					double sumX = dX[0], sumY = dY[0];
					double best = sumX * sumX + sumY * sumY;
					bestX = sumX;
					bestY = sumY;
					double endAngle = ang[end];
					const double window = T.oriWindow;
					while (start != total) {
						const double startAngle = ang[start];
						while (angleDist(startAngle, endAngle) <= window) {
							sumX += dX[end];
							sumY += dY[end];
							const double mag = sumX * sumX + sumY * sumY;
							if (mag > best) { best = mag; bestX = sumX; bestY = sumY; }
							end++;
							if (end >= total) end = 0;
							endAngle = ang[end];
							if (end == start) break;
						}
						sumX -= dX[start];
						sumY -= dY[start];
						start++;
					}
				}
				bestX = __shfl(bestX, 0, 64);
				bestY = __shfl(bestY, 0, 64);
			}
			angle = atan2(bestY, bestX);
			dirX = bestX; dirY = bestY;
		} else {
			double Dx = 0, Dy = 0;
			if (lane == 0) {
				for (int i = 0; i < n; i++) { Dx += dX[i]; Dy += dY[i]; }
			}
			Dx = __shfl(Dx, 0, 64);
			Dy = __shfl(Dy, 0, 64);
			angle = atan2(Dy, Dx);
			dirX = Dx; dirY = Dy;
		}
		waveSync();
	}
	DSTAMP(3);
	if (lane == 0 && P.angles) P.angles[g] = angle;
	if (!P.desc) return;

	// ------------------------------------------------------------------ descriptor
	// c = cos(angle), s = sin(angle) (DescribePointSurf.describe :190-191).  When the angle was formed here as atan2(dirY, dirX), the unit
	// vector dir / |dir| is cos / sin of the exact angle to within 2 ulp -- as close to the reference's cos(fl(atan2)) as two libm's are to
	// each other ("last ulp of Math.sin/cos/atan2") -- for a tenth of the instructions of an fp64 sincos.
	double c, s;
	{
		const double h2 = dirX * dirX + dirY * dirY;
		if (h2 > 1.0e-200 && h2 < 1.0e200) {   // wave-uniform; false for a caller-supplied angle (dir = 0)
			const double h = sqrt(h2);
			c = dirX / h; s = dirY / h;
		} else {
			c = cos(angle); s = sin(angle);
		}
	}
	const double scale = kp.scale;
	const int descR = gradRadius(T.widthSample * scale);
	const int widthLargeGrid = CFG ? 4 : T.widthLargeGrid, widthSubRegion = CFG ? 5 : T.widthSubRegion;
	const bool stableDesc = CFG == 1 ? true : CFG == 2 ? false : (T.stable != 0);
	const int regionSize = widthLargeGrid * widthSubRegion;
	const int regionR = regionSize / 2;
	const int overLap = CFG == 1 ? 2 : CFG == 2 ? 0 : (T.stable ? T.overLap : 0);
	const int gridW = regionSize + 2 * overLap;
	const int nsamp = gridW * gridW;
	// The samples wait in LDS for the sums below.  The default configurations keep them in the tap type (4 bytes: fp32 gradients, or the exact
	// integer box differences of a GrayS32 integral image) and widen them when they are read -- the same value either way -- so that the
	// descriptor phase fits the LDS the orientation phase needs (four workgroups per CU); the generic instantiation stores doubles.
	typedef typename std::conditional<CFG != 0, TAP, double>::type SampT;
	SampT* sX = (SampT*)lds;
	SampT* sY = sX + nsamp;
	double* feat = (double*)(sY + nsamp);   // 8-byte aligned: nsamp is even for the default grids
	// Laplacian sign (computeLaplaceSign): kernelDerivXX(9s) + kernelDerivYY(9s) at the rounded location = four clamped box sums of
	// four corners each.  Lane t < 16 fetches corner (t & 3) of box (t >> 2) now; the sign is assembled at the end of the kernel, so
	// the scattered loads are hidden behind the descriptor work.
	TAP lapTap = TAP(0);
	if (P.white && lane < 16) {
		const int x = (int)(kp.x + 0.5), y = (int)(kp.y + 0.5);
		const int si = (int)ceil(scale);
		const int size = 9 * si;
		const int blockW = size / 3, blockH = size - blockW - 1;
		const int r1 = blockW / 2, r2 = blockW + r1, r3 = blockH / 2;
		const int box = lane >> 2, corner = lane & 3;
		const int rx = box == 0 ? r2 : box == 1 ? r1 : r3;   // half extents of the box along x / y
		const int ry = box == 0 ? r3 : box == 1 ? r3 : box == 2 ? r2 : r1;
		int bx0 = x - rx - 1, by0 = y - ry - 1, bx1 = x + rx, by1 = y + ry;
		bx0 = min(bx0, W - 1); by0 = min(by0, H - 1); bx1 = min(bx1, W - 1); by1 = min(by1, H - 1);
		// corner order of block_zero: br, tr, bl, tl
		const int cx = (corner == 0 || corner == 1) ? bx1 : bx0;
		const int cy = (corner == 0 || corner == 2) ? by1 : by0;
		if (cx >= 0 && cy >= 0) lapTap = d[(long long)cy * stride + cx];
	}
	const int dof = CFG ? 64 : T.dof;
	const int nb = P.nBands > 0 ? P.nBands : 1;
#pragma unroll 1
	for (int band = 0; band < nb; band++) {
	const TAP* __restrict__ db = P.nBands > 0 ? (const TAP*)P.bandData + (long long)img * P.bandImageStride + (long long)band * P.bandStride : d;
	if (band > 0) waveSync();   // the previous band's sums have read sX, sY
	const TapBases<TAP> G = makeBases<TAP>(db, descR, stride, W, H, P.kps);
	{
		// sample grid in 8x8 blocks: the 64 lanes of one pass cover a compact (8 scale)^2 patch of the image, so a wave-level gather
		// touches a few dozen cache lines instead of up to 64.  The LDS layout stays [iy][ix].
		const double c_x = kp.x + 0.5, c_y = kp.y + 0.5;
		const int blocksPerSide = (gridW + 7) >> 3;
		const int nblocks = blocksPerSide * blocksPerSide;
		const int ly = lane >> 3, lx = lane & 7;
		const int gridOff = regionR + overLap;
		if (CFG != 0) {
			// default grids (24 or 20 samples per side) are three blocks per side: one batch = one row of blocks.  The products of the
			// reference's expressions  pixelX = (int)(c_x + c*regionX - s*regionY),  pixelY = (int)(c_y + s*regionX + c*regionY)
			// (DescribePointSurfMod.java:144-150, DescribePointSurf.java:256-262; evaluated left to right) depend on the sample's column only
			// (first two terms) or row only (last term), so they are formed once per lane and block column / block row -- same values,
			// same roundings, a third of the fp64 instructions.
			static_assert(DESC_BATCH == 3, "one batch = one row of three blocks");
			double PX[3], PY[3];
#pragma unroll
			for (int u = 0; u < 3; u++) {
				const int rX = 8 * u + lx - gridOff;
				const double regionX = rX * scale;
				PX[u] = c_x + c * regionX;
				PY[u] = c_y + s * regionX;
			}
#pragma unroll 1
			for (int by = 0; by < 3; by++) {
				GradTaps<TAP> tp[3];
				const int iy = 8 * by + ly;
				const double regionY = (iy - gridOff) * scale;
				const double qx = s * regionY, qy = c * regionY;
#pragma unroll
				for (int u = 0; u < 3; u++) {
					const bool on = iy < gridW && 8 * u + lx < gridW;
					gradFetch<TAP>(G, (int)(PX[u] - qx), (int)(PY[u] + qy), on, tp[u]);
				}
#pragma unroll
				for (int u = 0; u < 3; u++) {
					TAP gx, gy;
					gradFinish<TAP>(tp[u], gx, gy);
					const int ix = 8 * u + lx;
					if (iy < gridW && ix < gridW) { sX[iy * gridW + ix] = (SampT)gx; sY[iy * gridW + ix] = (SampT)gy; }
				}
			}
		} else {
#pragma unroll 1
		for (int b0 = 0; b0 < nblocks; b0 += DESC_BATCH) {
			GradTaps<TAP> tp[DESC_BATCH];
			int at[DESC_BATCH];
#pragma unroll
			for (int u = 0; u < DESC_BATCH; u++) {
				const int b = b0 + u;
				const int by = b / blocksPerSide, bx = b - by * blocksPerSide;
				const int iy = 8 * by + ly, ix = 8 * bx + lx;
				const bool on = b < nblocks && iy < gridW && ix < gridW;
				at[u] = on ? iy * gridW + ix : -1;
				const int rY = iy - gridOff, rX = ix - gridOff;
				const double regionY = rY * scale;
				const double regionX = rX * scale;
				const int pixelX = (int)(c_x + c * regionX - s * regionY);
				const int pixelY = (int)(c_y + s * regionX + c * regionY);
				gradFetch<TAP>(G, pixelX, pixelY, on, tp[u]);
			}
#pragma unroll
			for (int u = 0; u < DESC_BATCH; u++) {
				TAP gx, gy;
				gradFinish<TAP>(tp[u], gx, gy);
				if (at[u] >= 0) { sX[at[u]] = (SampT)gx; sY[at[u]] = (SampT)gy; }
			}
		}
		}
	}
	waveSync();
	DSTAMP(4);
	const int T_w = widthSubRegion + 2 * overLap;  // samples per sub-region side (<= TWT, enforced on the host)
	if (widthLargeGrid == 4) {
		// 16 sub-regions x 4 lanes: lane (sub, q) takes every fourth sample of its sub-region (t = q, q+4, ...), builds the rotated gradient
		// once and adds it into all four sums; the four partial sums meet in two butterfly steps.  A third of the instructions of one lane
		// per output (which evaluates w*gx, w*gy and the rotation four times over).  The order of the additions differs from the reference's
		// single running sum per output: a few ulp of the sum, ~1e-16 on the unit descriptor, against the 1e-5 bar.
		const int sub = lane >> 2, q = lane & 3;
		const int suby = sub >> 2, subx = sub & 3;
		const int rY = -regionR + suby * widthSubRegion, rX = -regionR + subx * widthSubRegion;
		const int base = (rY + regionR) * gridW + rX + regionR;
		const int nT = T_w * T_w;
		const double ms = -s;
		double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll 3
		for (int t = q; t < nT; t += 4) {
			const int i = t / T_w, j = t - i * T_w;
			const double vx = (double)sX[base + i * gridW + j], vy = (double)sY[base + i * gridW + j];
			const double w = stableDesc ? T.weightSub[t] : T.weightFast[(regionR + rY + i) * regionSize + regionR + rX + j];
			const double dx = w * vx, dy = w * vy;
			const double pdx = c * dx + s * dy;
			const double pdy = ms * dx + c * dy;
			a0 += pdx; a1 += fabs(pdx); a2 += pdy; a3 += fabs(pdy);
		}
#pragma unroll
		for (int o = 1; o <= 2; o <<= 1) {
			a0 += __shfl_xor(a0, o, 64); a1 += __shfl_xor(a1, o, 64); a2 += __shfl_xor(a2, o, 64); a3 += __shfl_xor(a3, o, 64);
		}
		double sum = q == 0 ? a0 : q == 1 ? a1 : q == 2 ? a2 : a3;
		if (stableDesc) sum = T.weightGrid[sub] * sum;
		feat[band * dof + lane] = sum;
	} else
#pragma unroll 1
	for (int f = lane; f < dof; f += 64) {
		const int sub = f >> 2, comp = f & 3;
		const int suby = sub / widthLargeGrid, subx = sub - suby * widthLargeGrid;
		const int rY = -regionR + suby * widthSubRegion, rX = -regionR + subx * widthSubRegion;
		// this lane's component as one linear form: pdx = c*dx + s*dy (comp 0,1), pdy = -s*dx + c*dy (comp 2,3); (-s)*dx is the exact
		// negation of s*dx, so the sum below is bit-identical to the reference's expression.  |.| for the odd components is a sign mask.
		const double cA = comp < 2 ? c : -s, cB = comp < 2 ? s : c;
		const unsigned long long absMask = (comp & 1) ? 0x7fffffffffffffffull : 0xffffffffffffffffull;
		double sum = 0;
#pragma unroll 1
		for (int i = 0; i < T_w; i++) {
			const int index = (rY + regionR + i) * gridW + rX + regionR;
			// one wait per row: the row's samples and weights are fetched together, then summed in the reference's order
			double vx[TWT], vy[TWT];
			double w[TWT];
#pragma unroll
			for (int j = 0; j < TWT; j++) {
				const bool on = j < T_w;
				vx[j] = on ? (double)sX[index + j] : 0.0;
				vy[j] = on ? (double)sY[index + j] : 0.0;
				w[j] = !on ? 0.0 : stableDesc ? T.weightSub[i * T_w + j] : T.weightFast[(regionR + rY + i) * regionSize + regionR + rX + j];
			}
#pragma unroll
			for (int j = 0; j < TWT; j++) {
				if (j < T_w) {
					const double dx = w[j] * vx[j];
					const double dy = w[j] * vy[j];
					const double v = cA * dx + cB * dy;
					sum += __longlong_as_double((long long)((unsigned long long)__double_as_longlong(v) & absMask));
				}
			}
		}
		if (stableDesc) sum = T.weightGrid[sub] * sum;
		feat[band * dof + f] = sum;
	}
	}   // bands
	waveSync();
	DSTAMP(5);
	// normalizeL2: sum of squares as a wave reduction (the reference adds the squares sequentially; the two orders differ by a few
	// ulp of the norm, ~1e-16 relative on the descriptor, against the 1e-5 bar)
	const int dofAll = dof * nb;
	double norm = 0;
	for (int f = lane; f < dofAll; f += 64) { const double v = feat[f]; norm += v * v; }
#pragma unroll
	for (int o = 32; o >= 1; o >>= 1) norm += __shfl_xor(norm, o, 64);
	double* out = P.desc + g * dofAll;
	if (norm == 0) {
		for (int f = lane; f < dofAll; f += 64) out[f] = feat[f];
	} else {
		norm = sqrt(norm);
		for (int f = lane; f < dofAll; f += 64) out[f] = feat[f] / norm;
	}
	if (P.white) {
		// block_zero = br - tr - bl + tl per box, then xx = 0 + b0*1 + b1*(-3), yy likewise, lap = (double)xx + (double)yy
		TAP bs[4];
#pragma unroll
		for (int b = 0; b < 4; b++) {
			const TAP br = __shfl(lapTap, 4 * b, 64), tr = __shfl(lapTap, 4 * b + 1, 64), bl = __shfl(lapTap, 4 * b + 2, 64), tl = __shfl(lapTap, 4 * b + 3, 64);
			bs[b] = br - tr - bl + tl;
		}
		// convolveSparse: ret = 0; ret += block * scale in the integral image's type (float scales for GrayF32, int for GrayS32), then widened
		TAP xx = 0;
		xx += bs[0] * TAP(1);
		xx += bs[1] * TAP(-3);
		TAP yy = 0;
		yy += bs[2] * TAP(1);
		yy += bs[3] * TAP(-3);
		double lap = (double)xx;
		lap += (double)yy;
		if (lane == 0) P.white[g] = lap > 0 ? 1 : 0;
	}
	DSTAMP(6);
}

// 1 = FactoryDetectDescribe.surfStable defaults, 2 = surfFast defaults (the CFG template argument of k_describe), 0 = anything else
static int bhip_describe_default_cfg(const SurfTables& t) {
	const bool defGrid = t.widthLargeGrid == 4 && t.widthSubRegion == 5 && t.dof == 64;
	if (defGrid && t.stable && t.overLap == 2 && t.oriStable && t.oriWidth == 17) return 1;
	if (defGrid && !t.stable && !t.oriStable && t.oriWidth == 13) return 2;
	return 0;
}

int bhip_describe_lds_bytes(const SurfTables& t, int nBands) {
	const int n = t.oriWidth * t.oriWidth;
	// merge-sort path: 28n while sorting, then dX dY sA Esched + the fp32 copy of the sorted angles (32n); counting-sort path: dX dY sA Esched
	// (28n) + the 1024 packed bin counters / offsets (2 KB, 16-byte aligned)
	const int ori = std::max(n * 32 + 16, ((28 * n + 15) & ~15) + 2048);
	const int overLap = t.stable ? t.overLap : 0;
	const int gridW = t.widthLargeGrid * t.widthSubRegion + 2 * overLap;
	const int ns = gridW * gridW;
	// sX sY | feat; the default configurations store 4-byte samples (see k_describe)
	const int sampBytes = bhip_describe_default_cfg(t) ? 4 : 8;
	const int desc = 2 * ns * sampBytes + t.dof * 8 * (nBands > 0 ? nBands : 1);
	int b = ori > desc ? ori : desc;
	return (b + 15) & ~15;
}

int bhip_launch_describe_ex(bhip_ctx* ctx, ImgView ii, const KeyPoint* kps, int cap, const int* imageStart, int batch, int singleImage, long long total,
							SurfTables t, const double* anglesIn, double* angles, double* desc, uint8_t* white, const int* perm, const DescPlanar* planar) {
	if (total <= 0) return BHIP_OK;
	DescParams P;
	P.ii = ii; P.kps = kps; P.cap = cap; P.imageStart = imageStart; P.batch = batch; P.singleImage = singleImage; P.total = total; P.t = t;
	P.anglesIn = anglesIn; P.angles = angles; P.desc = desc; P.white = white; P.perm = perm;
	P.bandData = nullptr; P.bandImageStride = P.bandStride = 0; P.nBands = 0; P.oriRadiusFactor = 2.0;
	if (planar) {
		if (planar->nBands > 0) {
			if (!planar->data) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad planar description request");
			P.bandData = planar->data; P.bandImageStride = planar->imageStride; P.bandStride = planar->bandStride; P.nBands = planar->nBands;
		}
		P.oriRadiusFactor = planar->oriRadiusFactor;
	}
	P.ldsPerWave = bhip_describe_lds_bytes(t, P.nBands);
	P.stamps = nullptr;
	P.serialOnly = bhip_env_flag("BHIP_DESCRIBE_SERIAL") ? 1 : 0;   // parity cross-checks of the two window sweeps / the two sort keys
	P.sort64 = bhip_env_flag("BHIP_DESCRIBE_SORT64") ? 1 : 0;
#ifdef BHIP_EXPERIMENTS
	{ const char* e = getenv("BHIP_DESCRIBE_STOP"); P.stopAfter = e ? atoi(e) : -1; }
	{ const char* e = getenv("BHIP_DESCRIBE_WSTOP"); P.windowStop = e ? atoi(e) : -1; }
#endif
	if (t.oriWidth * t.oriWidth > 64 * ORI_EPL_MAX) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "orientation sample grid too large for the GPU path");
	if (t.widthSubRegion + 2 * (t.stable ? t.overLap : 0) > DESC_ROW_MAX) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "SURF sub-region too wide for the GPU path");
	if (P.ldsPerWave * 4 > 160 * 1024) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "orientation/descriptor sample grid too large for LDS");
	const long long blocks = ((((total + 3) / 4) + 7) / 8) * 8;   // whole rounds over the 8 XCDs
	if (blocks > 0x7fffffffLL) return bhip_fail(ctx, BHIP_ERR_INVALID, "too many key points");
	if ((long long)ii.stride * ii.height > 0x3fffffffLL || ii.stride >= (1 << 17)) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "integral image too large for 32-bit tap offsets");
#ifdef BHIP_EXPERIMENTS
	const char* stampPath = getenv("BHIP_DESCRIBE_STAMPS");
	if (stampPath && total > 1000 && !(planar && planar->intTaps)) {
		// diagnostic build: phase shares of the describe kernel (never a quoted run time)
		unsigned long long* dev = nullptr;
		if (hipMalloc(&dev, (size_t)total * 128) == hipSuccess) {
			(void)hipMemsetAsync(dev, 0, (size_t)total * 128, ctx->stream);
			P.stamps = dev;
			{
				const int epl = (t.oriWidth * t.oriWidth + 63) / 64;
				const int tw = t.widthSubRegion + 2 * (t.stable ? t.overLap : 0);
				if (bhip_describe_default_cfg(t) == 1) hipLaunchKernelGGL((k_describe<true, 5, 9, float, 1>), dim3((unsigned)blocks), dim3(256), (size_t)P.ldsPerWave * 4, ctx->stream, P);
				else hipLaunchKernelGGL((k_describe<true, 8, 16, float>), dim3((unsigned)blocks), dim3(256), (size_t)P.ldsPerWave * 4, ctx->stream, P);
			}
			std::vector<unsigned long long> h((size_t)total * 16);
			(void)hipMemcpyAsync(h.data(), dev, (size_t)total * 128, hipMemcpyDeviceToHost, ctx->stream);
			(void)hipStreamSynchronize(ctx->stream);
			(void)hipFree(dev);
			double sum[7] = {0}, wsum[6] = {0};
			long long cnt = 0;
			for (long long k = 0; k < total; k++) {
				const unsigned long long* tt = &h[(size_t)k * 16];
				if (!tt[6] || !tt[0]) continue;
				for (int i = 1; i <= 6; i++) sum[i] += (double)(tt[i] - tt[i - 1]);
				if (tt[8] && tt[12]) { wsum[0] += (double)(tt[8] - tt[2]); for (int i = 1; i <= 4; i++) wsum[i] += (double)(tt[8 + i] - tt[7 + i]); wsum[5] += (double)(tt[3] - tt[12]); }
				cnt++;
			}
			FILE* f = fopen(stampPath, "a");
			if (f && cnt) {
				fprintf(f, "waves %lld  avg cycles: samples %.0f  sort %.0f  window+atan2 %.0f  descSamples %.0f  sums %.0f  norm+laplace %.0f\n", cnt,
						sum[1] / cnt, sum[2] / cnt, sum[3] / cnt, sum[4] / cnt, sum[5] / cnt, sum[6] / cnt);
			}
			if (f && cnt) fprintf(f, "   window split: c(a) %.0f  scan+validate %.0f  prefix %.0f  candidates %.0f  reduce %.0f  atan2+rest %.0f\n", wsum[0] / cnt, wsum[1] / cnt,
								  wsum[2] / cnt, wsum[3] / cnt, wsum[4] / cnt, wsum[5] / cnt);
			if (f) fclose(f);
			BHIP_HIP(ctx, hipGetLastError());
			return BHIP_OK;
		}
	}
#endif
	{
		// per key point: every orientation and descriptor sample reads 12 integral-image taps; angle + descriptor + sign written once
		const int gridWv = t.widthLargeGrid * t.widthSubRegion + 2 * (t.stable ? t.overLap : 0);
		const int nbv = P.nBands > 0 ? P.nBands : 1;
		const double perKp = (double)(t.oriWidth * t.oriWidth + nbv * gridWv * gridWv) * 12 * 4 + 8.0 * t.dof * nbv + 8 + 1;
		ProfScope ps(ctx, "k_describe", perKp * (double)total);
		const int epl = (t.oriWidth * t.oriWidth + 63) / 64;
		const int tw = t.widthSubRegion + 2 * (t.stable ? t.overLap : 0);
		const dim3 grid((unsigned)blocks), block(256);
		size_t ldsBytes = (size_t)P.ldsPerWave * 4;
#ifdef BHIP_EXPERIMENTS
		{ const char* e = getenv("BHIP_DESCRIBE_LDSPAD"); if (e) ldsBytes += (size_t)atoi(e); }   // occupancy experiments only
#endif
		// default configurations get compile-time sizes (CFG 1 / 2); anything else runs the generic instantiations
		const bool cfgStable = bhip_describe_default_cfg(t) == 1, cfgFast = bhip_describe_default_cfg(t) == 2;
		const bool ints = planar && planar->intTaps;   // GrayS32 integral image(s)
		const void* fn;
		if (cfgStable) fn = ints ? (const void*)k_describe<false, 5, 9, int, 1> : (const void*)k_describe<false, 5, 9, float, 1>;
		else if (cfgFast) fn = ints ? (const void*)k_describe<false, 3, 5, int, 2> : (const void*)k_describe<false, 3, 5, float, 2>;
		else if (epl == 5 && tw == 9) fn = ints ? (const void*)k_describe<false, 5, 9, int> : (const void*)k_describe<false, 5, 9, float>;
		else if (epl == 3 && tw == 5) fn = ints ? (const void*)k_describe<false, 3, 5, int> : (const void*)k_describe<false, 3, 5, float>;
		else fn = ints ? (const void*)k_describe<false, 8, 16, int> : (const void*)k_describe<false, 8, 16, float>;
		// colour SURF with many bands / large sample grids need more than the default 64 KB of dynamic LDS
		if (ldsBytes > 65536) BHIP_HIP(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
		void* args[] = {(void*)&P};
#ifdef BHIP_EXPERIMENTS
		{
			// experiment: orientation and descriptor as two launches of the same kernel (how long does each half take on its own?)
			const char* e = getenv("BHIP_DESCRIBE_SPLIT");
			if (e && e[0] == '1' && P.angles && P.desc && !P.anglesIn) {
				DescParams Po = P, Pd = P;
				Po.desc = nullptr; Po.white = nullptr;
				Pd.anglesIn = P.angles;
				void* ao[] = {(void*)&Po};
				void* ad[] = {(void*)&Pd};
				{ ProfScope p1(ctx, "k_describe_ori"); BHIP_HIP(ctx, hipLaunchKernel(fn, grid, block, ao, ldsBytes, ctx->stream)); }
				{ ProfScope p2(ctx, "k_describe_desc"); BHIP_HIP(ctx, hipLaunchKernel(fn, grid, block, ad, ldsBytes, ctx->stream)); }
				return BHIP_OK;
			}
			if (e && e[0] == '2' && P.angles && P.desc && !P.anglesIn) {
				// experiment: both halves at the same time on two streams (the descriptor half reads the angles the previous call left in the
				// buffer: timing only, results are stale) -- how much of the fused kernel's time is lost to the two halves not overlapping?
				static hipStream_t side = nullptr;
				static hipEvent_t evFork = nullptr, evJoin = nullptr;
				if (!side) { (void)hipStreamCreateWithFlags(&side, hipStreamNonBlocking); (void)hipEventCreateWithFlags(&evFork, hipEventDisableTiming); (void)hipEventCreateWithFlags(&evJoin, hipEventDisableTiming); }
				DescParams Po = P, Pd = P;
				Po.desc = nullptr; Po.white = nullptr;
				static double* staleAngles = nullptr; static long long staleN = 0;
				if (staleN < P.total) { if (staleAngles) (void)hipFree(staleAngles); (void)hipMalloc(&staleAngles, (size_t)P.total * 8); (void)hipMemsetAsync(staleAngles, 0, (size_t)P.total * 8, ctx->stream); staleN = P.total; }
				Pd.anglesIn = staleAngles; Pd.angles = nullptr;
				void* ao[] = {(void*)&Po};
				void* ad[] = {(void*)&Pd};
				BHIP_HIP(ctx, hipEventRecord(evFork, ctx->stream));
				BHIP_HIP(ctx, hipStreamWaitEvent(side, evFork, 0));
				BHIP_HIP(ctx, hipLaunchKernel(fn, grid, block, ad, ldsBytes, side));
				BHIP_HIP(ctx, hipLaunchKernel(fn, grid, block, ao, ldsBytes, ctx->stream));
				BHIP_HIP(ctx, hipEventRecord(evJoin, side));
				BHIP_HIP(ctx, hipStreamWaitEvent(ctx->stream, evJoin, 0));
				BHIP_HIP(ctx, hipMemcpyAsync(staleAngles, P.angles, (size_t)P.total * 8, hipMemcpyDeviceToDevice, ctx->stream));
				return BHIP_OK;
			}
		}
#endif
		BHIP_HIP(ctx, hipLaunchKernel(fn, grid, block, args, ldsBytes, ctx->stream));
	}
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}
