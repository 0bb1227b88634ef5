// K3: strict block non-maximum suppression + 3x3x3 scale-space test + sub-pixel fit + ordered compaction.
//
// Reference:
//   NonMaxBlock.process                       F:alg/feature/detect/extract/NonMaxBlock.java:69-94
//   NonMaxBlockSearchStrict.Max.searchBlock   F:alg/feature/detect/extract/NonMaxBlockSearchStrict.java:56-79, checkLocalMax :196-221
//   FastHessianFeatureDetector.findLocalScaleSpaceMax :230-298, checkMax :304-313, polyPeak :336-350
//       (F:alg/feature/detect/interest/FastHessianFeatureDetector.java)
//
// The block algorithm accepts, per (r+1)^2 block, the first strict block maximum iff it is >= threshold, != Float.MAX_VALUE and no other
// pixel of its (2r+1)^2 neighbourhood (clamped to the image) is >= it.  A pixel passing the neighbourhood test is necessarily the unique
// maximum of its block, so the accepted set is "all strict (2r+1)^2 maxima inside the ignore border" and the reference's output order
// (USE_CONCURRENT=false) is block-raster order.  Each pixel is tested by one thread; an accepted pixel sets bit (blockY*nbx+blockX) of a
// per-image bitmap, which makes the ordering a popcount-prefix: rank = #bits below.  No sort, no atomics on the ordering path.
// fp32 compares only, so keypoint indices are bit-exact whenever the intensity image is.
#include "hessian_dev.h"
#include <cfloat>

// strict-maximum test of NonMaxBlockSearchStrict.Max for pixel (x,y) with value v.  The window is read without early exits so the
// loads are independent (a branch per load turns the window into a chain of dependent memory round trips).
template <int RT>   // RT > 0: compile-time radius, fully unrolled; RT == 0: run-time radius r
__device__ __forceinline__ bool strictLocalMax(const float* __restrict__ img, int stride, int w, int h, int x, int y, int r, float v, float thr) {
	if (!(v >= thr) || v == FLT_MAX) return false;
	bool isMax = true;
	if (RT > 0) {
		float nb[(2 * RT + 1) * (2 * RT + 1)];
#pragma unroll
		for (int j = -RT; j <= RT; j++)
#pragma unroll
			for (int i = -RT; i <= RT; i++) {
				const int xx = x + i, yy = y + j;
				const bool in = xx >= 0 && xx < w && yy >= 0 && yy < h && !(i == 0 && j == 0);
				nb[(j + RT) * (2 * RT + 1) + i + RT] = in ? img[(long long)yy * stride + xx] : -INFINITY;
			}
#pragma unroll
		for (int k = 0; k < (2 * RT + 1) * (2 * RT + 1); k++)
			if (nb[k] >= v) isMax = false;
	} else {
		const int x0 = max(x - r, 0), x1 = min(x + r, w - 1), y0 = max(y - r, 0), y1 = min(y + r, h - 1);
		for (int j = y0; j <= y1; j++) {
			const float* row = img + (long long)j * stride;
			for (int i = x0; i <= x1; i++)
				if (row[i] >= v && !(i == x && j == y)) isMax = false;
		}
	}
	return isMax;
}

__device__ __forceinline__ float polyPeak(float lower, float middle, float upper) {
	const float a = 0.5f * lower - middle + 0.5f * upper;
	const float b = 0.5f * upper - 0.5f * lower;
	if (a == 0.0f) return 0.0f;
	return -b / (2.0f * a);
}

// FastHessianFeatureDetector.findLocalScaleSpaceMax :268-294 for one NMS maximum (x,y) of the middle level with value val, given the 3x3
// neighbourhoods lo[] / up[] of the lower and upper level (row-major, centre at [4]): checkMax on both, polyPeak fits.
__device__ __forceinline__ bool scaleSpaceFit(const float* lo, const float* up, const float* __restrict__ mid, int stride, const DetectLevelParams& p, int x,
											   int y, float val, KeyPoint& kp) {
	bool below = true;
#pragma unroll
	for (int k = 0; k < 9; k++)
		if (lo[k] >= val || up[k] >= val) below = false;
	if (!below) return false;
	const float peakX = polyPeak(mid[(long long)y * stride + x - 1], val, mid[(long long)y * stride + x + 1]);
	const float peakY = polyPeak(mid[(long long)(y - 1) * stride + x], val, mid[(long long)(y + 1) * stride + x]);
	const float peakS = polyPeak(lo[4], val, up[4]);
	const float interpX = ((float)x + peakX) * (float)p.skip;
	const float interpY = ((float)y + peakY) * (float)p.skip;
	const float interpS = (float)p.sizeMid + peakS * (float)(p.sizeMid - p.sizeLower);
	kp.x = (double)interpX;
	kp.y = (double)interpY;
	kp.scale = 1.2 * (double)interpS / 9.0;
	return true;
}

// An outer level of an octave (first / last kernel size) is only ever read around the NMS maxima of its neighbour level, so the
// stand-alone path may leave it uncomputed: its 3x3 values are then evaluated here, from the integral image, exactly as k_hessian would.
struct VirtualLevel {
	HessLevel L;       // geometry on this octave's lattice
	int on;            // 0: the level is in memory
};
struct NmsVirtual {
	ImgView ii;
	int intTaps;
	VirtualLevel lower, upper;
};
__device__ __forceinline__ void neighbourhood9(const float* __restrict__ lvl, int stride, int w, int h, int x, int y, float* out) {
	// 0 outside the image (never hit: candidates hugging the ignore border were dropped)
#pragma unroll
	for (int j = -1; j <= 1; j++)
#pragma unroll
		for (int i = -1; i <= 1; i++) {
			const int xx = x + i, yy = y + j;
			out[(j + 1) * 3 + i + 1] = (xx >= 0 && xx < w && yy >= 0 && yy < h) ? lvl[(long long)yy * stride + xx] : 0.0f;
		}
}
// border guard + scaleSpaceFit on levels held in memory.  Fills kp.x / kp.y / kp.scale and returns true when the point is a key point.
__device__ __forceinline__ bool scaleSpaceKeyPoint(const float* __restrict__ lower, const float* __restrict__ mid, const float* __restrict__ upper, int stride,
													const DetectLevelParams& p, int r, int x, int y, float val, KeyPoint& kp) {
	const int w = p.w, h = p.h;
	// candidates hugging the ignore border are dropped
	const int ignoreR = p.border + r;
	if (x < ignoreR || x >= w - ignoreR || y < ignoreR || y >= h - ignoreR) return false;
	float lo[9], up[9];
	neighbourhood9(lower, stride, w, h, x, y, lo);
	neighbourhood9(upper, stride, w, h, x, y, up);
	return scaleSpaceFit(lo, up, mid, stride, p, x, y, val, kp);
}

struct NmsParams {
	const float* lower;
	const float* mid;
	const float* upper;
	long long imageStride;
	int stride;
	DetectLevelParams p;
	int radius;
	float threshold;
	unsigned int* bitmap;
	int bitmapWords;
	KeyPoint* cand;
	int* candCount;
	int cap;
	int listOnly;
	NmsVirtual virt;
};

#define VSURV 64      // maxima per pass of k_nms_scalespace that wait for an evaluated outer level (such a pass tests 64 candidates)
#define NMS_ROWS 8   // rows per thread: the column neighbours are shared and the grid has 8x fewer, longer-lived blocks (4: 0.63 ms, 8: 0.49 ms, 16: 0.48 ms)
__device__ __forceinline__ void emitKeyPoint(const NmsParams& P, int img, int x, int y, KeyPoint kp) {
	const int b = P.p.border, step = P.radius + 1;
	const unsigned int bit = P.p.bitBase + (unsigned)((y - b) / step) * (unsigned)P.p.nbx + (unsigned)((x - b) / step);
	atomicOr(&P.bitmap[(long long)img * P.bitmapWords + (bit >> 5)], 1u << (bit & 31));
	const int slot = atomicAdd(&P.candCount[img], 1);
	if (slot < P.cap) {
		kp.key = bit;
		kp.pad = 0;
		P.cand[(long long)img * P.cap + slot] = kp;
	}
}

__global__ __launch_bounds__(256) void k_nms_scalespace(NmsParams P) {
	const int b = P.p.border;
	const int w = P.p.w, h = P.p.h;
	const int x = b + blockIdx.x * blockDim.x + threadIdx.x;
	const int yBase = b + blockIdx.y * NMS_ROWS;
	const int img = blockIdx.z;
	const float* mid = P.mid + (long long)img * P.imageStride;
	const int stride = P.stride;
	const int r = P.radius;
	// column values for rows yBase-1 .. yBase+NMS_ROWS and the left / right neighbours of the NMS_ROWS centre rows, all independent loads
	float col[NMS_ROWS + 2], lf[NMS_ROWS], rt[NMS_ROWS];
#pragma unroll
	for (int k = 0; k < NMS_ROWS + 2; k++) {
		const int yy = yBase - 1 + k;
		col[k] = (yy >= 0 && yy < h && x < w - b) ? mid[(long long)yy * stride + x] : -INFINITY;
	}
#pragma unroll
	for (int k = 0; k < NMS_ROWS; k++) {
		const int yy = yBase + k;
		const bool rowIn = yy < h && x < w - b;
		lf[k] = (rowIn && x >= 1) ? mid[(long long)yy * stride + x - 1] : -INFINITY;
		rt[k] = (rowIn && x + 1 < w) ? mid[(long long)yy * stride + x + 1] : -INFINITY;
	}
	// Survivors of the cheap tests are rare (a few per cent) but almost every wave has one, so running the full-window test in place
	// would make every wave walk the expensive path with one or two active lanes.  They are compacted into a block-wide list first
	// and then tested densely, one candidate per thread.  (The output order is fixed later by the bitmap, not by emission order.)
	__shared__ int candList[256 * NMS_ROWS];
	__shared__ int candCount;
	if (threadIdx.x == 0) candCount = 0;
	__syncthreads();
	if (x < w - b) {
#pragma unroll
		for (int k = 0; k < NMS_ROWS; k++) {
			const int y = yBase + k;
			const float val = col[k + 1];
			const bool pass = y < h - b && val >= P.threshold && val != FLT_MAX &&
							  // most pixels above the threshold lose against a direct neighbour (r >= 1, so the four are inside the strict-max window)
							  !(lf[k] >= val || rt[k] >= val || col[k] >= val || col[k + 2] >= val);
			if (pass) candList[atomicAdd(&candCount, 1)] = (k << 16) | (int)threadIdx.x;
		}
	}
	__syncthreads();
	const int ncand = candCount;
	const bool anyVirtual = P.virt.lower.on || P.virt.upper.on;   // launch-uniform
	// maxima that still need an outer level evaluated: (x, y) and, per evaluated level, the nine responses (one thread per response)
	__shared__ int survXY[VSURV];
	__shared__ float survVal[VSURV][18];
	__shared__ int survCount;
	const int passN = anyVirtual ? VSURV : (int)blockDim.x;   // candidates per pass: a pass with an evaluated level keeps at most VSURV survivors in LDS
	for (int c0 = 0; c0 < ncand; c0 += passN) {   // workgroup-uniform trip count
		if (anyVirtual) {
			if (threadIdx.x == 0) survCount = 0;
			__syncthreads();
		}
		const int ci = c0 + threadIdx.x;
		if ((int)threadIdx.x < passN && ci < ncand) {
			const int code = candList[ci];
			const int x = b + blockIdx.x * blockDim.x + (code & 0xFFFF);
			const int y = yBase + (code >> 16);
			const float val = mid[(long long)y * stride + x];
			if (r == 2 ? strictLocalMax<2>(mid, stride, w, h, x, y, r, val, P.threshold) : strictLocalMax<0>(mid, stride, w, h, x, y, r, val, P.threshold)) {
				if (P.listOnly) {
					// maxFeaturesPerScale > 0: every NMS maximum is listed with its intensity; selection and the scale-space test follow in k_select_nbest
					KeyPoint kp;
					kp.x = (double)x; kp.y = (double)y; kp.scale = (double)val;
					emitKeyPoint(P, img, x, y, kp);
				} else if (!anyVirtual) {
					KeyPoint kp;
					if (scaleSpaceKeyPoint(P.lower + (long long)img * P.imageStride, mid, P.upper + (long long)img * P.imageStride, stride, P.p, r, x, y, val, kp))
						emitKeyPoint(P, img, x, y, kp);
				} else {
					const int ignoreR = b + r;
					if (!(x < ignoreR || x >= w - ignoreR || y < ignoreR || y >= h - ignoreR)) {
						// the level held in memory first: most maxima lose against it and never reach the evaluated level
						float nb[9];
						bool alive = true;
						if (!P.virt.lower.on || !P.virt.upper.on) {
							neighbourhood9(!P.virt.lower.on ? P.lower + (long long)img * P.imageStride : P.upper + (long long)img * P.imageStride, stride, w, h, x, y, nb);
#pragma unroll
							for (int k = 0; k < 9; k++)
								if (nb[k] >= val) alive = false;
						}
						if (alive) {
							const int s = atomicAdd(&survCount, 1);   // <= passN = VSURV per pass
							survXY[s] = (y << 16) | x;
							if (!P.virt.lower.on || !P.virt.upper.on) {
								const int o = !P.virt.lower.on ? 0 : 9;
#pragma unroll
								for (int k = 0; k < 9; k++) survVal[s][o + k] = nb[k];
							}
						}
					}
				}
			}
		}
		if (anyVirtual) {
			__syncthreads();
			const int ns = survCount;
			const int per = (P.virt.lower.on ? 9 : 0) + (P.virt.upper.on ? 9 : 0);
			for (int t = threadIdx.x; t < ns * per; t += blockDim.x) {
				const int s = t / per;
				int k = t - s * per;
				const bool up = !P.virt.lower.on || k >= 9;
				if (k >= 9) k -= 9;
				const int xy = survXY[s];
				const int xx = (xy & 0xFFFF) + k % 3 - 1, yy = (xy >> 16) + k / 3 - 1;
				const HessLevel& L = up ? P.virt.upper.L : P.virt.lower.L;
				const float* base = P.virt.ii.data + (long long)img * P.virt.ii.imageStride;
				float v = 0.0f;   // 0 outside the image (never hit: maxima hugging the ignore border were dropped)
				if (xx >= 0 && xx < w && yy >= 0 && yy < h)
					v = P.virt.intTaps ? hessianCompute<int>((const int*)base, P.virt.ii.stride, P.virt.ii.width, P.virt.ii.height, L, P.p.skip, w, h, xx, yy)
									   : hessianCompute<float>(base, P.virt.ii.stride, P.virt.ii.width, P.virt.ii.height, L, P.p.skip, w, h, xx, yy);
				survVal[s][(up ? 9 : 0) + k] = v;
			}
			__syncthreads();
			for (int s = threadIdx.x; s < ns; s += blockDim.x) {
				const int xy = survXY[s];
				const int x = xy & 0xFFFF, y = xy >> 16;
				KeyPoint kp;
				if (scaleSpaceFit(&survVal[s][0], &survVal[s][9], mid, stride, P.p, x, y, mid[(long long)y * stride + x], kp)) emitKeyPoint(P, img, x, y, kp);
			}
		}
	}
}

int bhip_launch_nms_scalespace(bhip_ctx* ctx, const float* lower, const float* mid, const float* upper, long long imageStride, int stride, int batch,
							   DetectLevelParams p, int radius, float threshold, unsigned int* bitmap, int bitmapWords, KeyPoint* cand, int* candCount,
							   int cap, bool listOnly, const ImgView* ii, bool intTaps) {
	const int rw = p.w - 2 * p.border, rh = p.h - 2 * p.border;
	if (rw <= 0 || rh <= 0) return BHIP_OK;
	NmsParams P{lower, mid, upper, imageStride, stride, p, radius, threshold, bitmap, bitmapWords, cand, candCount, cap, listOnly ? 1 : 0, {}};
	P.virt.lower.on = P.virt.upper.on = 0;
	P.virt.intTaps = intTaps ? 1 : 0;
	if (!lower || !upper) {
		// an outer level that was not computed: evaluated on demand from the integral image
		if (!ii || listOnly) return bhip_fail(ctx, BHIP_ERR_INVALID, "a level that is not in memory needs the integral image");
		P.virt.ii = *ii;
		if (!lower) { P.virt.lower.on = 1; P.virt.lower.L = bhipMakeHessLevel(p.sizeLower, p.skip); }
		if (!upper) { P.virt.upper.on = 1; P.virt.upper.L = bhipMakeHessLevel(p.sizeUpper, p.skip); }
	}
	dim3 grid((rw + 255) / 256, (rh + NMS_ROWS - 1) / NMS_ROWS, batch);
	{
		ProfScope ps(ctx, "k_nms_scalespace", 4.0 * p.w * p.h * batch);  // the mid level read once
		hipLaunchKernelGGL(k_nms_scalespace, grid, dim3(256), 0, ctx->stream, P);
	}
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}

// ---- exclusive prefix of per-word popcounts: one workgroup per image ----
__global__ __launch_bounds__(1024) void k_word_prefix(const unsigned int* __restrict__ bitmap, int words, unsigned int* __restrict__ prefix, int* __restrict__ totals) {
	__shared__ unsigned int waveSum[16];
	__shared__ unsigned int carry;
	const int img = blockIdx.x;
	const unsigned int* bm = bitmap + (long long)img * words;
	unsigned int* pf = prefix + (long long)img * words;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	if (threadIdx.x == 0) carry = 0;
	__syncthreads();
	for (int base = 0; base < words; base += 1024) {
		const int i = base + threadIdx.x;
		const unsigned int c = i < words ? __popc(bm[i]) : 0u;
		// inclusive scan inside the wave
		unsigned int s = c;
#pragma unroll
		for (int o = 1; o < 64; o <<= 1) {
			const unsigned int t = __shfl_up(s, o, 64);
			if (lane >= o) s += t;
		}
		if (lane == 63) waveSum[wave] = s;
		__syncthreads();
		unsigned int off = carry;
		for (int k = 0; k < wave; k++) off += waveSum[k];
		if (i < words) pf[i] = off + s - c;
		__syncthreads();
		if (threadIdx.x == 1023) carry = off + s;
		__syncthreads();
	}
	if (threadIdx.x == 0 && totals) totals[img] = (int)carry;
}

int bhip_launch_word_prefix(bhip_ctx* ctx, const unsigned int* bitmap, int bitmapWords, int batch, unsigned int* wordPrefix, int* totals) {
	if (batch <= 0 || bitmapWords <= 0) return BHIP_OK;
	{
		ProfScope ps(ctx, "k_word_prefix", 8.0 * bitmapWords * batch);
		hipLaunchKernelGGL(k_word_prefix, dim3(batch), dim3(1024), 0, ctx->stream, bitmap, bitmapWords, wordPrefix, totals);
	}
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}

// ---- scatter the unordered candidates to their block-raster rank ----
__global__ __launch_bounds__(256) void k_rank_scatter(const unsigned int* __restrict__ bitmap, int words, const unsigned int* __restrict__ prefix,
													   const KeyPoint* __restrict__ cand, const int* __restrict__ candCount, int cap,
													   KeyPoint* __restrict__ sorted) {
	const int img = blockIdx.y;
	const int n = min(candCount[img], cap);
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const KeyPoint kp = cand[(long long)img * cap + i];
	const unsigned int word = kp.key >> 5, bit = kp.key & 31;
	const unsigned int below = bitmap[(long long)img * words + word] & ((1u << bit) - 1u);
	const unsigned int rank = prefix[(long long)img * words + word] + __popc(below);
	sorted[(long long)img * cap + rank] = kp;
}

int bhip_launch_rank_scatter(bhip_ctx* ctx, const unsigned int* bitmap, int bitmapWords, unsigned int* wordPrefix, const KeyPoint* cand,
							 const int* candCount, int cap, int batch, KeyPoint* sorted) {
	if (batch <= 0 || cap <= 0) return BHIP_OK;
	dim3 grid((cap + 255) / 256, batch);
	{
		ProfScope ps(ctx, "k_rank_scatter");
		hipLaunchKernelGGL(k_rank_scatter, grid, dim3(256), 0, ctx->stream, bitmap, bitmapWords, (const unsigned int*)wordPrefix, cand, candCount, cap, sorted);
	}
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// maxFeaturesPerScale > 0   (FastHessianFeatureDetector.java:255-262 -> SelectNBestFeatures.process :51-93)
//
// The NMS maxima of every middle level have been listed in block-raster order with their intensities (k_nms_scalespace in listOnly mode
// + the popcount ranking).  One wave per (image, level):
//   1. n <= N: the list is kept as it is.  Otherwise keys = -intensity and QuickSelect.selectIndex(keys, N, n, indexes) is run by lane 0
//      exactly as the sequential routine runs (ddogleg's routine = Numerical Recipes `select` with an index array; see the oracle's
//      quickSelectIndex -- the exchange sequence decides the output order and which of several equal keys survive, so it is not parallelised);
//   2. the first min(n, N) entries, in that order, go through the border guard, the 3x3x3 scale-space test and the sub-pixel fit, 64 at a
//      time, and the survivors are appended in order (ballot + popcount).
// The results of a level are written at the level's own offset in the NMS list (they are never more than the list itself), then
// k_compact_levels closes the gaps.
// ---------------------------------------------------------------------------------------------------------------
#define SEL_LDS 4096   // lists up to this long are selected in LDS, longer ones in their global scratch

struct SelectParams {
	const float* lower;
	const float* mid;
	const float* upper;
	long long imageStride;
	int stride;
	DetectLevelParams p;
	int radius;
	int target;
	const unsigned int* bitmap;
	const unsigned int* prefix;
	int bitmapWords;
	const KeyPoint* nms;   // [image][cap] ranked NMS maxima: x, y = pixel, scale = intensity
	int cap;
	float* keyBuf;         // [image][cap]
	int* idxBuf;           // [image][cap]
	KeyPoint* out;         // [image][cap]
	int* levelStart;       // [image][nlv]
	int* levelCount;       // [image][nlv]
	int levelIndex, nlv;
};

__device__ __forceinline__ int bitRank(const unsigned int* __restrict__ bm, const unsigned int* __restrict__ pf, unsigned int bit) {
	return (int)(pf[bit >> 5] + __popc(bm[bit >> 5] & ((1u << (bit & 31)) - 1u)));
}

__device__ void quickSelectIndexSeq(float* data, int k, int n, int* indexes) {
	int l = 0, ir = n - 1;
#define QS_SWAP(a_, b_) do { const float tf = data[a_]; data[a_] = data[b_]; data[b_] = tf; const int ti = indexes[a_]; indexes[a_] = indexes[b_]; indexes[b_] = ti; } while (0)
	for (;;) {
		if (ir <= l + 1) {
			if (ir == l + 1 && data[ir] < data[l]) QS_SWAP(l, ir);
			return;
		}
		const int mid = (l + ir) >> 1, lp1 = l + 1;
		QS_SWAP(mid, lp1);
		if (data[l] > data[ir]) QS_SWAP(l, ir);
		if (data[lp1] > data[ir]) QS_SWAP(lp1, ir);
		if (data[l] > data[lp1]) QS_SWAP(l, lp1);
		int i = lp1, j = ir;
		const float a = data[lp1];
		const int indexA = indexes[lp1];
		for (;;) {
			do i++; while (data[i] < a);   // data[ir] >= a and data[l] <= a are the sentinels, as in the sequential routine
			do j--; while (data[j] > a);
			if (j < i) break;
			QS_SWAP(i, j);
		}
		data[lp1] = data[j]; data[j] = a;
		indexes[lp1] = indexes[j]; indexes[j] = indexA;
		if (j >= k) ir = j - 1;
		if (j <= k) l = i;
	}
#undef QS_SWAP
}

__global__ __launch_bounds__(64) void k_select_nbest(SelectParams P) {
	__shared__ float ldsKey[SEL_LDS];
	__shared__ int ldsIdx[SEL_LDS];
	const int img = blockIdx.x;
	const int lane = threadIdx.x;
	const unsigned int* bm = P.bitmap + (long long)img * P.bitmapWords;
	const unsigned int* pf = P.prefix + (long long)img * P.bitmapWords;
	const int start = bitRank(bm, pf, P.p.bitBase);
	const int end = bitRank(bm, pf, P.p.bitBase + (unsigned)P.p.nbx * (unsigned)P.p.nby);
	const int n = end - start;
	const KeyPoint* list = P.nms + (long long)img * P.cap + start;
	KeyPoint* out = P.out + (long long)img * P.cap + start;
	const bool select = n > P.target;
	const int m = select ? P.target : n;
	const bool inLds = n <= SEL_LDS;
	float* key = inLds ? ldsKey : P.keyBuf + (long long)img * P.cap + start;
	int* idx = inLds ? ldsIdx : P.idxBuf + (long long)img * P.cap + start;
	if (select) {
		for (int i = lane; i < n; i += 64) {
			key[i] = -(float)list[i].scale;   // intensity was widened from float: exact
			idx[i] = i;
		}
		__threadfence_block();
		__builtin_amdgcn_wave_barrier();
		if (lane == 0) quickSelectIndexSeq(key, P.target, n, idx);
		__threadfence_block();
		__builtin_amdgcn_wave_barrier();
	}
	const float* lower = P.lower + (long long)img * P.imageStride;
	const float* mid = P.mid + (long long)img * P.imageStride;
	const float* upper = P.upper + (long long)img * P.imageStride;
	int base = 0;
	for (int j0 = 0; j0 < m; j0 += 64) {
		const int j = j0 + lane;
		bool ok = false;
		KeyPoint kp;
		kp.x = kp.y = kp.scale = 0; kp.key = 0; kp.pad = 0;
		if (j < m) {
			const KeyPoint src = list[select ? idx[j] : j];
			const int x = (int)src.x, y = (int)src.y;
			ok = scaleSpaceKeyPoint(lower, mid, upper, P.stride, P.p, P.radius, x, y, (float)src.scale, kp);
		}
		const unsigned long long vote = __ballot(ok);
		if (ok) out[base + __popcll(vote & ((1ull << lane) - 1ull))] = kp;
		base += __popcll(vote);
	}
	if (lane == 0) {
		P.levelStart[(long long)img * P.nlv + P.levelIndex] = start;
		P.levelCount[(long long)img * P.nlv + P.levelIndex] = base;
	}
}

// one wave per image: the per-level results, in level order, are packed into `dst` and the image's key point count is written
__global__ __launch_bounds__(64) void k_compact_levels(const KeyPoint* __restrict__ src, int cap, const int* __restrict__ levelStart, const int* __restrict__ levelCount,
														int nlv, KeyPoint* __restrict__ dst, int* __restrict__ totals) {
	const int img = blockIdx.x, lane = threadIdx.x;
	int off = 0;
	for (int l = 0; l < nlv; l++) {
		const int s = levelStart[(long long)img * nlv + l], c = levelCount[(long long)img * nlv + l];
		for (int i = lane; i < c; i += 64) dst[(long long)img * cap + off + i] = src[(long long)img * cap + s + i];
		off += c;
	}
	if (lane == 0) totals[img] = off;
}

int bhip_launch_select_nbest(bhip_ctx* ctx, const float* lower, const float* mid, const float* upper, long long imageStride, int stride, int batch,
							 DetectLevelParams p, int radius, int target, const unsigned int* bitmap, const unsigned int* prefix, int bitmapWords,
							 const KeyPoint* nms, int cap, float* keyBuf, int* idxBuf, KeyPoint* out, int* levelStart, int* levelCount, int levelIndex, int nlv) {
	if (batch <= 0) return BHIP_OK;
	SelectParams P{lower, mid, upper, imageStride, stride, p, radius, target, bitmap, prefix, bitmapWords, nms, cap, keyBuf, idxBuf, out, levelStart, levelCount,
				   levelIndex, nlv};
	{
		ProfScope ps(ctx, "k_select_nbest");
		hipLaunchKernelGGL(k_select_nbest, dim3(batch), dim3(64), 0, ctx->stream, P);
	}
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}

int bhip_launch_compact_levels(bhip_ctx* ctx, const KeyPoint* src, int cap, const int* levelStart, const int* levelCount, int nlv, int batch, KeyPoint* dst,
							   int* totals) {
	if (batch <= 0) return BHIP_OK;
	{
		ProfScope ps(ctx, "k_compact_levels");
		hipLaunchKernelGGL(k_compact_levels, dim3(batch), dim3(64), 0, ctx->stream, src, cap, levelStart, levelCount, nlv, dst, totals);
	}
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}

// stand-alone SelectNBestFeatures.process(intensity, corners, positive) for n > target: one wave; keys = -/+ intensity(x,y), the
// sequential QuickSelect, then the first `target` points in index order
__global__ __launch_bounds__(64) void k_select_nbest_xy(const float* __restrict__ img, int stride, const int16_t* __restrict__ xy, int n, int target, int positive,
														 float* key, int* idx, int16_t* __restrict__ out) {
	const int lane = threadIdx.x;
	for (int i = lane; i < n; i += 64) {
		const float v = img[(long long)xy[2 * i + 1] * stride + xy[2 * i]];
		key[i] = positive ? -v : v;
		idx[i] = i;
	}
	__threadfence_block();
	__builtin_amdgcn_wave_barrier();
	if (lane == 0) quickSelectIndexSeq(key, target, n, idx);
	__threadfence_block();
	__builtin_amdgcn_wave_barrier();
	for (int i = lane; i < target; i += 64) {
		out[2 * i] = xy[2 * idx[i]];
		out[2 * i + 1] = xy[2 * idx[i] + 1];
	}
}

int bhip_launch_select_nbest_xy(bhip_ctx* ctx, const float* img, int stride, const int16_t* xy, int n, int target, bool positive, float* key, int* idx,
								int16_t* out) {
	hipLaunchKernelGGL(k_select_nbest_xy, dim3(1), dim3(64), 0, ctx->stream, img, stride, xy, n, target, positive ? 1 : 0, key, idx, out);
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}

// ---- stand-alone NMS over a batch of intensity images (BOverrideFactoryFeatureExtractor.nonmax, GeneralFeatureDetector, config-5 chain) ----
// Same structure as k_nms_scalespace: NMS_ROWS rows per thread, a cheap four-neighbour test, survivors compacted per block and tested
// densely.  An accepted pixel sets the bit of its (r+1)^2 block (at most one pixel per block can pass) and records its position inside the
// block, so the ordered list is rebuilt from the bitmap alone (k_blocks_to_xy) without a second pass over the image.
struct NonmaxParams {
	const float* img;
	long long imageStride;
	int stride, w, h, radius, border, nbx, bitmapWords;
	float threshold;
	unsigned int* bitmap;          // [batch][bitmapWords]
	unsigned short* posInBlock;    // [batch][nbx*nby]
	long long blocksPerImage;
};
__global__ __launch_bounds__(256) void k_nonmax_blocks(NonmaxParams P) {
	const int b = P.border, w = P.w, h = P.h, r = P.radius, stride = P.stride;
	const int x = b + blockIdx.x * blockDim.x + threadIdx.x;
	const int yBase = b + blockIdx.y * NMS_ROWS;
	const float* img = P.img + (long long)blockIdx.z * P.imageStride;
	float col[NMS_ROWS + 2], lf[NMS_ROWS], rt[NMS_ROWS];
#pragma unroll
	for (int k = 0; k < NMS_ROWS + 2; k++) {
		const int yy = yBase - 1 + k;
		col[k] = (yy >= 0 && yy < h && x < w - b) ? img[(long long)yy * stride + x] : -INFINITY;
	}
#pragma unroll
	for (int k = 0; k < NMS_ROWS; k++) {
		const int yy = yBase + k;
		const bool rowIn = yy < h && x < w - b;
		lf[k] = (rowIn && x >= 1) ? img[(long long)yy * stride + x - 1] : -INFINITY;
		rt[k] = (rowIn && x + 1 < w) ? img[(long long)yy * stride + x + 1] : -INFINITY;
	}
	__shared__ int candList[256 * NMS_ROWS];
	__shared__ int candCount;
	if (threadIdx.x == 0) candCount = 0;
	__syncthreads();
	if (x < w - b) {
#pragma unroll
		for (int k = 0; k < NMS_ROWS; k++) {
			const int y = yBase + k;
			const float val = col[k + 1];
			const bool pass = y < h - b && val >= P.threshold && val != FLT_MAX && !(lf[k] >= val || rt[k] >= val || col[k] >= val || col[k + 2] >= val);
			if (pass) candList[atomicAdd(&candCount, 1)] = (k << 16) | (int)threadIdx.x;
		}
	}
	__syncthreads();
	const int ncand = candCount;
	const int step = r + 1;
	for (int ci = threadIdx.x; ci < ncand; ci += blockDim.x) {
		const int code = candList[ci];
		const int cx = b + blockIdx.x * blockDim.x + (code & 0xFFFF);
		const int cy = yBase + (code >> 16);
		const float val = img[(long long)cy * stride + cx];
		if (!(r == 2 ? strictLocalMax<2>(img, stride, w, h, cx, cy, r, val, P.threshold) : strictLocalMax<0>(img, stride, w, h, cx, cy, r, val, P.threshold))) continue;
		const int bx = (cx - b) / step, by = (cy - b) / step;
		const unsigned int bit = (unsigned)by * (unsigned)P.nbx + (unsigned)bx;
		atomicOr(&P.bitmap[(long long)blockIdx.z * P.bitmapWords + (bit >> 5)], 1u << (bit & 31));
		P.posInBlock[(long long)blockIdx.z * P.blocksPerImage + bit] = (unsigned short)((cy - b - by * step) * step + (cx - b - bx * step));
	}
}

int bhip_launch_nonmax_blocks(bhip_ctx* ctx, const float* img, long long imageStride, int stride, int w, int h, int batch, int radius, float threshold, int border,
							  unsigned int* bitmap, int bitmapWords, unsigned short* posInBlock, int nbx, int nby) {
	const int rw = w - 2 * border, rh = h - 2 * border;
	if (rw <= 0 || rh <= 0 || batch <= 0) return BHIP_OK;
	if (radius > 254) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "NMS radius too large");
	NonmaxParams P{img, imageStride, stride, w, h, radius, border, nbx, bitmapWords, threshold, bitmap, posInBlock, (long long)nbx * nby};
	dim3 grid((rw + 255) / 256, (rh + NMS_ROWS - 1) / NMS_ROWS, batch);
	{
		ProfScope ps(ctx, "k_nonmax_blocks", 4.0 * w * h * batch);
		hipLaunchKernelGGL(k_nonmax_blocks, grid, dim3(256), 0, ctx->stream, P);
	}
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}

// one thread per bitmap word: every set bit (= accepted block, ascending = block-raster order) writes (x,y) at its rank
__global__ __launch_bounds__(256) void k_blocks_to_xy(const unsigned int* __restrict__ bitmap, const unsigned int* __restrict__ prefix, int words,
													   const unsigned short* __restrict__ posInBlock, long long blocksPerImage, int nbx, int step, int border,
													   int16_t* __restrict__ xy, int cap) {
	const int word = blockIdx.x * blockDim.x + threadIdx.x;
	if (word >= words) return;
	const long long img = blockIdx.y;
	unsigned int bits = bitmap[img * words + word];
	unsigned int rank = prefix[img * words + word];
	while (bits) {
		const int bit = __ffs(bits) - 1;
		bits &= bits - 1;
		const long long idx = (long long)word * 32 + bit;
		const int by = (int)(idx / nbx), bx = (int)(idx - (long long)by * nbx);
		const int pos = posInBlock[img * blocksPerImage + idx];
		if ((int)rank < cap) {
			xy[(img * cap + rank) * 2] = (int16_t)(border + bx * step + pos % step);
			xy[(img * cap + rank) * 2 + 1] = (int16_t)(border + by * step + pos / step);
		}
		rank++;
	}
}

int bhip_launch_blocks_to_xy(bhip_ctx* ctx, const unsigned int* bitmap, const unsigned int* wordPrefix, int bitmapWords, const unsigned short* posInBlock,
							 int nbx, int nby, int batch, int radius, int border, int16_t* xy, int cap) {
	if (batch <= 0 || bitmapWords <= 0 || cap <= 0) return BHIP_OK;
	ProfScope ps(ctx, "k_blocks_to_xy");
	hipLaunchKernelGGL(k_blocks_to_xy, dim3((bitmapWords + 255) / 256, batch), dim3(256), 0, ctx->stream, bitmap, wordPrefix, bitmapWords, posInBlock,
					   (long long)nbx * nby, nbx, radius + 1, border, xy, cap);
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}


// ---------------------------------------------------------------------------------------------------------------
// Processing order for the describe stage.  Results stay in detection order; only the order in which key points are WORKED ON changes:
// image-major, then by coarse image tile (KPT x KPT tiles per image, raster), so the key points in flight on one XCD gather from one
// compact part of the integral image and their taps hit that XCD's L2.  perm[slot] = compact index of the key point handled by slot.
// ---------------------------------------------------------------------------------------------------------------
#define KPT 8
__global__ __launch_bounds__(256) void k_kp_tile_count(const KeyPoint* __restrict__ kps, int cap, const int* __restrict__ start, int W, int H,
													   int* __restrict__ hist) {
	const int img = blockIdx.y;
	const int n = start[img + 1] - start[img];
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const KeyPoint kp = kps[(long long)img * cap + i];
	const int tx = min(KPT - 1, max(0, (int)(kp.x * KPT / W))), ty = min(KPT - 1, max(0, (int)(kp.y * KPT / H)));
	atomicAdd(&hist[img * (KPT * KPT) + ty * KPT + tx], 1);
}
// one wave per image: exclusive scan of the 64 tile counts (KPT*KPT == 64), in place
__global__ __launch_bounds__(64) void k_kp_tile_scan(int* __restrict__ hist) {
	const int img = blockIdx.x, lane = threadIdx.x;
	const int c = hist[img * 64 + lane];
	int sc = c;
#pragma unroll
	for (int o = 1; o < 64; o <<= 1) {
		const int t = __shfl_up(sc, o, 64);
		if (lane >= o) sc += t;
	}
	hist[img * 64 + lane] = sc - c;
}
__global__ __launch_bounds__(256) void k_kp_tile_scatter(const KeyPoint* __restrict__ kps, int cap, const int* __restrict__ start, int W, int H,
														 int* __restrict__ cursor, int* __restrict__ perm) {
	const int img = blockIdx.y;
	const int s0 = start[img];
	const int n = start[img + 1] - s0;
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	const KeyPoint kp = kps[(long long)img * cap + i];
	const int tx = min(KPT - 1, max(0, (int)(kp.x * KPT / W))), ty = min(KPT - 1, max(0, (int)(kp.y * KPT / H)));
	const int pos = atomicAdd(&cursor[img * (KPT * KPT) + ty * KPT + tx], 1);
	perm[s0 + pos] = s0 + i;
}

int bhip_launch_kp_spatial_order(bhip_ctx* ctx, const KeyPoint* kps, int cap, const int* start, int batch, int maxCount, int W, int H, int* hist /*batch*64*/,
								 int* perm) {
	if (batch <= 0 || maxCount <= 0) return BHIP_OK;
	static_assert(KPT * KPT == 64, "k_kp_tile_scan assumes 64 tiles");
	BHIP_HIP(ctx, hipMemsetAsync(hist, 0, (size_t)batch * 64 * 4, ctx->stream));
	ProfScope ps(ctx, "k_kp_spatial_order", 0);
	dim3 grid((maxCount + 255) / 256, batch);
	hipLaunchKernelGGL(k_kp_tile_count, grid, dim3(256), 0, ctx->stream, kps, cap, start, W, H, hist);
	hipLaunchKernelGGL(k_kp_tile_scan, dim3(batch), dim3(64), 0, ctx->stream, hist);
	hipLaunchKernelGGL(k_kp_tile_scatter, grid, dim3(256), 0, ctx->stream, kps, cap, start, W, H, hist, perm);
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}
