// K6 (MFMA path): batched greedy L2 association whose results are bit-identical to the reference's fp64 sequential scores.
//
// Reference: AssociateGreedy.associate  F:alg/feature/associate/AssociateGreedy.java:65-118 with
//            DescriptorDistance.euclideanSq  F:alg/descriptor/DescriptorDistance.java:55-64  (SURVEY hard part 3)
//
// The matrix cores only FILTER: they find, for every row and every column, the short list of pairs that can attain the exact minimum; the
// listed pairs (about 1.15 per row / column on SURF descriptors) are then re-scored with the reference's sequential fp64 loop and the
// reference's rules are applied to those exact values.  Because the filter only needs a rigorous error bound, not accuracy, it runs in
// fp16 (v_mfma_f32_32x32x16_f16, 16x the fp32 matrix rate) on descriptors scaled by one power of two 2^-q per call so that the largest
// squared norm lies in [1/4, 1):
//     a^ = fl16(2^-q a),  n_a = |2^-q a|^2 (fp32, rounded up) = hi + lo (two fp16),
//     d~(i,j) = 1 + n_a + n_b - 2 <a^_i, b^_j> -- ONE K = 80 contraction: the row holds [-2 a^ | hi lo 1 1 1 | 0...], the column [b^ | 1 1 hi lo 1 | 0...]
//   (the constant 1 keeps every score a positive float, so minima are unsigned-integer minima of the bit patterns: v_min3_u32, no NaN
//    canonicalisation, LDS / global atomicMin on the raw bits)
// Error bound (u = 2^-11, s = 2^-25 fp16 subnormal half-spacing, |a'|,|b'| < 1 after scaling; products of fp16 are exact in fp32):
//     input rounding   2[(2u + u^2)|a'||b'| + 8s(1+u)(|a'|+|b'|) + 64 s^2]   <= 9.78e-4 (n_a + n_b) + 1e-6     (double rounding f64->f32->f16: + 2^-24 rel.)
//     norm split       2^-22 (n_a + n_b) + 2^-24 ;  norm round-up 4 * 2^-24 (n_a + n_b)
//     fp32 accumulate  85 additions, <= 2^-23 relative each even if truncating: 2^-16 (1 + n_a + n_b + 2|a'||b'|) <= 2^-15 (n_a + n_b) + 1.6e-5
//   =>  |d~ - 1 - d'| <= eps(i,j) = C16 (n_a + n_b) + ABS16,   C16 = 1.03e-3,  ABS16 = 2e-5   (d' = 2^-2q d, the exact scaled distance)
// Pass 1 reduces d~ to per-row and per-column minima.  Pass 2 recomputes the tiles (same instructions, same bits) and lists every pair with
//     d~(i,j) <= rowmin~(i) + band(i)   /   d~(i,j) <= colmin~(j) + band(j),     band = 2 (C16 (n + max n) + ABS16)   (>= 2 eps)
// The true row arg-min (and every exact tie of it) and the true column minimum (and every exact tie) are provably inside those lists and
// everything outside is strictly larger than the list's best.  Exact re-score + the reference's rules:
//   forward : smallest exact score, largest destination index among exact ties, inclusive maxFitError
//   backward: (i -> m) survives iff i is the only row attaining the exact minimum of column m
// Degenerate inputs (candidate list overflow, e.g. many near-equal descriptors; non-finite values) fall back to the exact VALU kernels.
//
// Layout: one 144-byte row per descriptor = 9 chunks of 8 halves (64 values + [hi lo 1 1 1 0 0 0]).  A workgroup keeps ONE column strip of
// a problem (<= 384 destination rows, fragment order) in LDS and walks its row range in wave tiles of 64 rows (two A tiles in VGPRs, the
// next tile's chunks requested one tile ahead): 20 MFMAs per 64-column step, every B fragment read from LDS feeds two of them, no barrier
// and no global traffic inside the sweep.  The norms ride in the contraction, so the epilogue is one v_min3_u32 per two scores (pass 1) or
// one compare per score (pass 2).  Column minima of the strip accumulate in LDS over all row tiles; row minima of a tile are reduced with
// DPP exchanges and meet the other strips in a global atomicMin.
// Bound: MFMA (fp16 matrix rate); algorithmic flops per pass = 2 * Ns * Nd * 64.
#include "common.h"
#include <cfloat>
#include <cmath>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

struct AssocProblem {
	int srcOff, ns, dstOff, nd;   // rows in the src / dst descriptor buffers
	int rowBase, colBase;         // compact offsets into the per-row / per-column work arrays
};
struct AssocBlock {
	int p, row0, row1, col0, col1;   // rows [row0, row1) x columns [col0, col1) of problem p
};
struct AssocCand {
	int p, i, j, pad;
	double score;
};

#define C16 1.03e-3f
#define ABS16 2.0e-5f
#define BIG16 60000.0f     // "norm" of a row / column that does not exist: its scores can never pass a threshold (real norms are < 1)
#define ROW_CHUNKS 9       // uint4 per stored row

// LDS writes of one lane visible to the other lanes of the same wave (lock-step wave: only the memory counter has to drain)
__device__ __forceinline__ void waveSyncLds() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// q with max norm * 2^-2q in [1/4, 1)  (0 when there is no positive norm)
__device__ __forceinline__ int scaleExp(float maxN) {
	if (!(maxN > 0.0f)) return 0;
	int e;
	(void)frexpf(maxN, &e);
	return (e + 1) >> 1;
}

// ---- squared norms (fp32, rounded up); flags[0] = non-finite seen, flags[1] = max norm (float bits, non-negative) ----
// 16 lanes per row, 4 consecutive values (two 16-byte loads) per lane; a lane group takes PREP_ROWS rows with all their loads in flight
#define PREP_ROWS 8
__global__ __launch_bounds__(256) void k_assoc_norms(const double* __restrict__ D, long long rows, float* __restrict__ nrm, int* __restrict__ flags) {
	const long long row0 = (((long long)blockIdx.x * 256 + threadIdx.x) >> 4) * PREP_ROWS;
	const int part = threadIdx.x & 15;
	if (row0 >= rows) return;   // whole 16-lane groups leave together
	double2 v0[PREP_ROWS], v1[PREP_ROWS];
#pragma unroll
	for (int q = 0; q < PREP_ROWS; q++) {
		const long long row = min(row0 + q, rows - 1);
		const double2* src = (const double2*)(D + row * 64 + 4 * part);
		v0[q] = src[0]; v1[q] = src[1];
	}
#pragma unroll
	for (int q = 0; q < PREP_ROWS; q++) {
		double s = (v0[q].x * v0[q].x + v0[q].y * v0[q].y) + (v1[q].x * v1[q].x + v1[q].y * v1[q].y);
#pragma unroll
		for (int o = 8; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
		if (part == 0 && row0 + q < rows) {
			const float n = (float)s;
			const float nUp = n * (1.0f + 4.0f * 5.9604644775390625e-08f);   // rounded up: a band computed from it can only grow
			nrm[row0 + q] = nUp;
			// one hot word for the whole launch: only touch it when this row actually raises the maximum (a handful of times per launch)
			if (!(s < 1e30)) atomicOr(&flags[0], 1);  // NaN, Inf or absurdly large: use the exact path
			else if (__float_as_int(nUp) > __hip_atomic_load(&flags[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&flags[1], __float_as_int(nUp));
		}
	}
}

// ---- fp64 -> scaled fp16 rows [64 values | hi lo 1 1 1 0 0 0]; nrm is rescaled in place ----
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_assoc_half(const double* __restrict__ D, long long rows, float* __restrict__ nrm, const int* __restrict__ flags,
													  _Float16* __restrict__ Hrow) {
	const long long row0 = (((long long)blockIdx.x * 256 + threadIdx.x) >> 4) * PREP_ROWS;
	const int part = threadIdx.x & 15;
	if (row0 >= rows) return;
	const int q_ = scaleExp(__int_as_float(flags[1]));
	double2 v0[PREP_ROWS], v1[PREP_ROWS];
#pragma unroll
	for (int q = 0; q < PREP_ROWS; q++) {
		const long long row = min(row0 + q, rows - 1);
		const double2* src = (const double2*)(D + row * 64 + 4 * part);
		v0[q] = src[0]; v1[q] = src[1];
	}
#pragma unroll
	for (int q = 0; q < PREP_ROWS; q++) {
		const long long row = row0 + q;
		if (row >= rows) break;
		_Float16* out = Hrow + row * (ROW_CHUNKS * 8);
		const h16x4 hv = {(_Float16)(float)ldexp(v0[q].x, -q_), (_Float16)(float)ldexp(v0[q].y, -q_), (_Float16)(float)ldexp(v1[q].x, -q_),
						  (_Float16)(float)ldexp(v1[q].y, -q_)};
		*(h16x4*)(out + 4 * part) = hv;
		if (part == 0) {
			const float n = ldexpf(nrm[row], -2 * q_);
			const _Float16 hi = (_Float16)n;
			const _Float16 lo = (_Float16)(n - (float)hi);
			const h16x8 e = {hi, lo, (_Float16)1.0f, (_Float16)1.0f, (_Float16)1.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
			*(h16x8*)(out + 64) = e;
			nrm[row] = n;
		}
	}
}

__global__ void k_fill_u32(unsigned int* p, long long n, unsigned int v) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) p[i] = v;
}

// thr = min~ + band, band = 2 (C16 (norm + maxNorm) + ABS16), scaled units
__global__ void k_assoc_thresholds(const AssocProblem* __restrict__ probs, int count, const float* __restrict__ nrmS, const float* __restrict__ nrmD,
								   const unsigned int* __restrict__ rowKey, const unsigned int* __restrict__ colKey, const int* __restrict__ flags,
								   float* __restrict__ rowThr, float* __restrict__ colThr) {
	const int p = blockIdx.y;
	const AssocProblem P = probs[p];
	const float maxRaw = __int_as_float(flags[1]);
	const float maxN = ldexpf(maxRaw, -2 * scaleExp(maxRaw));
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P.ns + P.nd; i += gridDim.x * blockDim.x) {
		if (i < P.ns) {
			const float band = 2.0f * (C16 * (nrmS[P.srcOff + i] + maxN) + ABS16);
			rowThr[P.rowBase + i] = __uint_as_float(rowKey[P.rowBase + i]) + band;   // still carries the +1 of the scores
		} else {
			const int j = i - P.ns;
			const float band = 2.0f * (C16 * (nrmD[P.dstOff + j] + maxN) + ABS16);
			colThr[P.colBase + j] = __uint_as_float(colKey[P.colBase + j]) + band;
		}
	}
}

struct MfmaArgs {
	const uint4* Hs;   // 9 chunks per row
	const uint4* Hd;
	const AssocProblem* probs;
	const AssocBlock* blocks;
	unsigned int* rowKey;
	unsigned int* colKey;
	const float* rowThr;
	const float* colThr;
	AssocCand* cand;
	int* counter;    // number of listed pairs
	int cap;
#ifdef BHIP_EXPERIMENTS
	int ablate;      // timing experiments only (BHIP_ASSOC_ABLATE): 1 no MFMA, 2 no per-step epilogue, 4 no A-fragment loads, 16 no row-minimum atomics, 32 no column-minimum flush
#endif
};

#define CAND_LDS 256   // per-wave staging slots in LDS (flushed between row tiles when half full; a 64-row tile lists about 25 pairs per strip)

// One list for both directions: a pair listed for its row also updates its column's exact minimum and vice versa, which is harmless (every
// listed pair is a real pair, and every pair that can attain a row / column minimum is listed).  Hits are rare per lane, so a hit lane
// appends with one LDS atomic; the wave flushes its staging area once, with ONE atomicAdd on the global counter (a single contended word
// sustains only ~88 atomics/us, MI355X_MICROARCH.md).  Entries that do not fit the staging area go to the global list directly.
// The staging area is full (a wave tile with several times the usual number of hits): straight to the global list.  Out of line: never on
// the hot path, and its 64-bit addressing must not cost the sweep registers.
__device__ __noinline__ void candSlow(AssocCand* cand, int* counter, int cap, int p, int rr, int col) {
	const int g = atomicAdd(counter, 1);
	if (g < cap) { AssocCand c; c.p = p; c.i = rr; c.j = col; c.pad = 0; c.score = 0; cand[g] = c; }
}
// one hit lane = one LDS atomic + one 8-byte LDS store.  Inlined at every accumulator register so that the pointers stay LDS pointers
// (ds_add_rtn / ds_write; as an out-of-line function it went through flat atomics and a full wait at entry: 0.08 ms of pass 2)
__device__ __forceinline__ void candAdd(const MfmaArgs& A, int2* buf, int* cnt, bool hit, int p, int rr, int col) {
	if (hit) {
		const int pos = atomicAdd(cnt, 1);
		if (pos < CAND_LDS) buf[pos] = make_int2(rr, col);
		else candSlow(A.cand, A.counter, A.cap, p, rr, col);
	}
}
__device__ __forceinline__ void candFlush(const MfmaArgs& A, const int2* buf, const int* cnt, int p, int lane) {
	waveSyncLds();
	const int n = min(*cnt, CAND_LDS);
	if (n == 0) return;
	int base = 0;
	if (lane == 0) base = atomicAdd(A.counter, n);
	base = __builtin_amdgcn_readfirstlane(base);
	for (int k = lane; k < n; k += 64) {
		if (base + k < A.cap) {
			const int2 v = buf[k];
			AssocCand c; c.p = p; c.i = v.x; c.j = v.y; c.pad = 0; c.score = 0;
			A.cand[base + k] = c;
		}
	}
}

__device__ __forceinline__ h16x8 asHalf8(uint4 v) { return __builtin_bit_cast(h16x8, v); }
__device__ __forceinline__ uint4 extChunk(float n0, float n1, float n2, float n3) {   // [n0 n1 n2 n3 1 0 0 0] as halves
	const h16x8 h = {(_Float16)n0, (_Float16)n1, (_Float16)n2, (_Float16)n3, (_Float16)1.0f, (_Float16)0.0f, (_Float16)0.0f, (_Float16)0.0f};
	return __builtin_bit_cast(uint4, h);
}

// B-stationary: a workgroup stages ONE column strip of a problem (<= STRIP_COLS destination rows, fragment order, 5.3 KB per 32 columns)
// in LDS and then walks its row range in wave tiles of 64 rows -- no barrier and no global traffic inside the sweep except the A fragments
// of the next tile.  The column minima of the strip accumulate in LDS over all row tiles and leave with one atomic per column.
#define STRIP_COLS 384
#define STRIP_TILES (STRIP_COLS / 32)
template <int PASS>
__global__ __launch_bounds__(256, 2) void k_assoc_mfma(MfmaArgs A) {
	// B tiles: [tile][chunk * 33 + column]; chunk 8 = [1 1 hi lo 1 ..], chunk 9 = zeros (k = 72..79, read by the upper half-wave)
	__shared__ uint4 tileB[STRIP_TILES][10 * 33];
	__shared__ unsigned int colLds[STRIP_COLS];       // PASS 1: column minima over this block's rows ; PASS 2: column thresholds (float bits)
	__shared__ float rowThrLds[PASS == 2 ? 4 : 1][64];   // PASS 2: thresholds of the wave's current 64 rows
	__shared__ int2 candLds[PASS == 2 ? 4 : 1][PASS == 2 ? CAND_LDS : 1];
	__shared__ int candCnt[4];
	const AssocBlock B = A.blocks[blockIdx.x];
	if (B.col0 >= B.col1) return;   // padding entry of the block table
	const AssocProblem P = A.probs[B.p];
	const int tid = threadIdx.x;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
	const int r = lane & 31, h = lane >> 5;
	const int ncol = B.col1 - B.col0;
	const int ntile = (ncol + 31) >> 5;
	int2* candBuf = candLds[PASS == 2 ? wave : 0];
	int* candN = &candCnt[wave];
	if (lane == 0) *candN = 0;

	// ---- stage the strip: consecutive threads fetch consecutive 16-byte chunks of a destination row (9 per row), four requests per
	// thread in flight before the first LDS store ----
	for (int g0 = tid; g0 < ntile * 32 * ROW_CHUNKS; g0 += 4 * 256) {
		uint4 v[4];
#pragma unroll
		for (int q = 0; q < 4; q++) {
			const int g = g0 + 256 * q;
			const int cl = g / ROW_CHUNKS, ch = g - cl * ROW_CHUNKS;
			if (g < ntile * 32 * ROW_CHUNKS && cl < ncol) {
				v[q] = A.Hd[(long long)(P.dstOff + B.col0 + cl) * ROW_CHUNKS + ch];
				if (ch == 8) { const unsigned int x = v[q].x; v[q].x = v[q].y; v[q].y = x; }   // [hi lo | 1 1 | 1 ..] -> [1 1 | hi lo | 1 ..]
			} else {
				v[q] = ch == 8 ? extChunk(1.0f, 1.0f, BIG16, 0.0f) : make_uint4(0, 0, 0, 0);   // a column that does not exist scores > BIG16
			}
		}
#pragma unroll
		for (int q = 0; q < 4; q++) {
			const int g = g0 + 256 * q;
			const int cl = g / ROW_CHUNKS, ch = g - cl * ROW_CHUNKS;
			if (g < ntile * 32 * ROW_CHUNKS) tileB[cl >> 5][ch * 33 + (cl & 31)] = v[q];
		}
	}
	for (int g = tid; g < ntile * 32; g += 256) tileB[g >> 5][9 * 33 + (g & 31)] = make_uint4(0, 0, 0, 0);
	for (int c = tid; c < ntile * 32; c += 256)
		colLds[c] = PASS == 1 ? 0xFFFFFFFFu : (c < ncol ? __float_as_uint(A.colThr[P.colBase + B.col0 + c]) : __float_as_uint(-INFINITY));
	__syncthreads();

	// ---- row tiles of this block's row range, 64 rows per wave and turn ----
	// A fragments: tile t, row rowT + 32 t + r; MFMA s (k = 16 s + 8 h ..) reads chunk 2 s + h, the fifth reads chunk 8 / zeros.  Raw chunks
	// of the NEXT row tile are requested before the current one is swept (their latency is of the order of a whole sweep).
	uint4 araw[2][5];
	auto loadA = [&](int rowT) {
#pragma unroll
		for (int t = 0; t < 2; t++) {
			const int row = rowT + 32 * t + r;
			if (row < B.row1) {
				const uint4* src = A.Hs + (long long)(P.srcOff + row) * ROW_CHUNKS;
#pragma unroll
				for (int s = 0; s < 4; s++) araw[t][s] = src[2 * s + h];
				araw[t][4] = h == 0 ? src[8] : make_uint4(0, 0, 0, 0);
			} else {
#pragma unroll
				for (int s = 0; s < 4; s++) araw[t][s] = make_uint4(0, 0, 0, 0);
				araw[t][4] = h == 0 ? extChunk(BIG16, 0.0f, 1.0f, 1.0f) : make_uint4(0, 0, 0, 0);
			}
		}
	};
	if (B.row0 + 64 * wave < B.row1) loadA(B.row0 + 64 * wave);
	for (int rowT = B.row0 + 64 * wave; rowT < B.row1; rowT += 256) {
		h16x8 a[2][5];
#pragma unroll
		for (int t = 0; t < 2; t++) {
#pragma unroll
			for (int s = 0; s < 4; s++) a[t][s] = asHalf8(araw[t][s]) * (_Float16)(-2.0f);
			a[t][4] = asHalf8(araw[t][4]);
		}
		if (rowT + 256 < B.row1 && !BHIP_ABLATE(A, 4)) loadA(rowT + 256);
		// per accumulator register: C layout row = (reg&3) + 8*(reg>>2) + 4h of the tile, col = lane&31
		unsigned int rowM[32];        // PASS 1: running row minimum of accumulator register (t, g) = [16 t + g] (bit pattern of a positive float)
#pragma unroll
		for (int k = 0; k < 32; k++) rowM[k] = 0xFFFFFFFFu;
		if (PASS == 2) {
			waveSyncLds();   // the previous tile's reads of rowThrLds are over
			rowThrLds[wave][lane] = rowT + lane < B.row1 ? A.rowThr[P.rowBase + rowT + lane] : -INFINITY;
			waveSyncLds();
		}
		for (int ct = 0; ct < ntile; ct += 2) {
			const bool two = ct + 1 < ntile;   // wave-uniform: an odd tile count leaves the last step with one column tile
			const int cA_ = 32 * ct + r, cB_ = cA_ + 32;
			f32x16 acc[2][2];
#pragma unroll
			for (int t = 0; t < 2; t++)
#pragma unroll
				for (int c = 0; c < 2; c++)
#pragma unroll
					for (int g = 0; g < 16; g++) acc[t][c][g] = (c == 0 || two) ? 0.0f : INFINITY;   // a missing second tile never wins a minimum / passes a threshold
#pragma unroll
			for (int s = 0; s < 5 && !BHIP_ABLATE(A, 1); s++) {
				const int ch = s < 4 ? 2 * s + h : 8 + h;
				const h16x8 b0 = asHalf8(tileB[ct][ch * 33 + r]);
				acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][s], b0, acc[0][0], 0, 0, 0);
				acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][s], b0, acc[1][0], 0, 0, 0);
				if (two) {
					const h16x8 b1 = asHalf8(tileB[ct + 1][ch * 33 + r]);
					acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0][s], b1, acc[0][1], 0, 0, 0);
					acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1][s], b1, acc[1][1], 0, 0, 0);
				}
			}
			if (BHIP_ABLATE(A, 2)) continue;
			if (PASS == 1) {
				unsigned int cA = 0xFFFFFFFFu, cB = 0xFFFFFFFFu;
#pragma unroll
				for (int t = 0; t < 2; t++)
#pragma unroll
					for (int g = 0; g < 16; g++) {
						const unsigned int uA = __float_as_uint(acc[t][0][g]), uB = __float_as_uint(acc[t][1][g]);   // positive floats (+inf for a missing tile): bit order == value order
						rowM[16 * t + g] = min(rowM[16 * t + g], min(uA, uB));
						cA = min(cA, uA);
						cB = min(cB, uB);
					}
				cA = min(cA, (unsigned int)__shfl_xor((int)cA, 32, 64));
				cB = min(cB, (unsigned int)__shfl_xor((int)cB, 32, 64));
				if (h == 0) {
					atomicMin(&colLds[cA_], cA);
					if (two) atomicMin(&colLds[cB_], cB);
				}
			} else {
				const float tA = __uint_as_float(colLds[cA_]);
				const float tB = two ? __uint_as_float(colLds[cB_]) : -INFINITY;
				const int colA = B.col0 + cA_, colB = B.col0 + cB_;
#pragma unroll
				for (int t = 0; t < 2; t++)
#pragma unroll
					for (int j = 0; j < 4; j++) {
						// rows 8 j + 4 h + 0..3 of tile t = accumulator registers 4 j .. 4 j + 3: one 16-byte LDS read (same address across a half-wave)
						const float4 rv = *(const float4*)&rowThrLds[wave][32 * t + 8 * j + 4 * h];
						const float rthr[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
						for (int q = 0; q < 4; q++) {
							const int g = 4 * j + q;
							const float dA = acc[t][0][g], dB = acc[t][1][g];
							const bool rA = dA <= rthr[q], rB = dB <= rthr[q], qA = dA <= tA, qB = dB <= tB;
							if (__ballot(rA || rB || qA || qB) != 0ull) {   // wave-uniform, a few times per step
								const int rr = rowT + 32 * t + (g & 3) + 8 * (g >> 2) + 4 * h;
								candAdd(A, candBuf, candN, rA || qA, B.p, rr, colA);
								candAdd(A, candBuf, candN, rB || qB, B.p, rr, colB);
							}
						}
					}
			}
		}
		if (PASS == 1) {
#pragma unroll
			for (int k = 0; k < 32; k++) {
				// minimum over the 32 lanes of the half-wave: four DPP exchanges inside a row of 16 lanes (quad swaps, half-row and row mirror:
				// plain VALU operands, no LDS round trip) and one cross-row exchange
				unsigned int v = rowM[k];
				if (!BHIP_ABLATE(A, 64)) {
					v = min(v, (unsigned int)__builtin_amdgcn_update_dpp((int)v, (int)v, 0xB1, 0xF, 0xF, false));    // quad_perm [1,0,3,2]
					v = min(v, (unsigned int)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x4E, 0xF, 0xF, false));    // quad_perm [2,3,0,1]
					v = min(v, (unsigned int)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x141, 0xF, 0xF, false));   // row_half_mirror
					v = min(v, (unsigned int)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x140, 0xF, 0xF, false));   // row_mirror
					v = min(v, (unsigned int)__shfl_xor((int)v, 16, 64));
				}
				const int rr = rowT + 32 * (k >> 4) + (k & 3) + 8 * ((k & 15) >> 2) + 4 * h;
				if (r == 0 && rr < B.row1 && !BHIP_ABLATE(A, 16)) atomicMin(&A.rowKey[P.rowBase + rr], v);
			}
		} else {
			waveSyncLds();
			if (*candN > CAND_LDS / 2) { candFlush(A, candBuf, candN, B.p, lane); waveSyncLds(); if (lane == 0) *candN = 0; waveSyncLds(); }
		}
	}
	if (PASS == 2) candFlush(A, candBuf, candN, B.p, lane);
	if (PASS == 1) {
		__syncthreads();   // every wave's LDS minima are in
		for (int c = tid; c < ncol && !BHIP_ABLATE(A, 32); c += 256) atomicMin(&A.colKey[P.colBase + B.col0 + c], colLds[c]);
	}
}

// exact sequential fp64 score of every listed pair + exact minima per row / column (scores are >= 0: bit order == value order)
__global__ __launch_bounds__(256) void k_assoc_exact(const double* __restrict__ src, const double* __restrict__ dst, const AssocProblem* __restrict__ probs,
													   AssocCand* __restrict__ cand, const int* __restrict__ counter, int cap,
													   unsigned long long* __restrict__ rowBest, unsigned long long* __restrict__ colBest) {
	const int n = min(*counter, cap);
	for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
		AssocCand* c = &cand[t];
		const AssocProblem P = probs[c->p];
		const double* a = src + (long long)(P.srcOff + c->i) * 64;
		const double* b = dst + (long long)(P.dstOff + c->j) * 64;
		double total = 0;
#pragma unroll 8
		for (int k = 0; k < 64; k++) {
			const double d = a[k] - b[k];
			total += d * d;
		}
		c->score = total;
		const unsigned long long bits = (unsigned long long)__double_as_longlong(total);
		atomicMin(&rowBest[P.rowBase + c->i], bits);
		atomicMin(&colBest[P.colBase + c->j], bits);
	}
}

// among the pairs that attain the exact minimum: largest destination index per row; count + any source index per column
__global__ __launch_bounds__(256) void k_assoc_argsel(const AssocProblem* __restrict__ probs, const AssocCand* __restrict__ cand,
														const int* __restrict__ counter, int cap, const unsigned long long* __restrict__ rowBest,
														const unsigned long long* __restrict__ colBest, int* __restrict__ rowArg, int* __restrict__ colArg,
														int* __restrict__ colCnt) {
	const int n = min(*counter, cap);
	for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
		const AssocCand c = cand[t];
		const AssocProblem P = probs[c.p];
		const unsigned long long bits = (unsigned long long)__double_as_longlong(c.score);
		if (bits == rowBest[P.rowBase + c.i]) atomicMax(&rowArg[P.rowBase + c.i], c.j);
		if (bits == colBest[P.colBase + c.j]) {
			atomicAdd(&colCnt[P.colBase + c.j], 1);
			atomicMax(&colArg[P.colBase + c.j], c.i);
		}
	}
}

__global__ __launch_bounds__(256) void k_assoc_resolve(const AssocProblem* __restrict__ probs, double maxErr, int backwards,
														 const unsigned long long* __restrict__ rowBest, const int* __restrict__ rowArg,
														 const int* __restrict__ colArg, const int* __restrict__ colCnt, int* __restrict__ pairs,
														 double* __restrict__ fit) {
	const AssocProblem P = probs[blockIdx.y];
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P.ns; i += gridDim.x * blockDim.x) {
		const int m = rowArg[P.rowBase + i];
		const double best = __longlong_as_double((long long)rowBest[P.rowBase + i]);
		int outPair = -1;
		double outFit = maxErr;
		if (m >= 0 && best <= maxErr) {
			outPair = m;
			outFit = best;
			if (backwards) {
				if (!(colArg[P.colBase + m] == i && colCnt[P.colBase + m] == 1)) { outPair = -1; outFit = DBL_MAX; }
			}
		}
		pairs[P.srcOff + i] = outPair;
		fit[P.srcOff + i] = outFit;
	}
}

// ---------------------------------------------------------------------------------------------------------------
// Returns BHIP_OK and sets *usedMfma = 1 when the MFMA path produced the result; *usedMfma = 0 means the caller must run the exact path.
int bhip_assoc_l2_mfma_batched(bhip_ctx* ctx, AssocMfmaWork& W, const double* dev_src, const double* dev_dst, int count, const long long* srcOff,
								 const int* ns, const long long* dstOff, const int* nd, double maxErr, int backwards, int* dev_pairs, double* dev_fit,
								 int* usedMfma) {
	*usedMfma = 0;
	if (count <= 0) { *usedMfma = 1; return BHIP_OK; }
	std::vector<AssocProblem> probs(count);
	std::vector<AssocBlock> blocks;
	long long maxSrcRow = 0, maxDstRow = 0, rowTotal = 0, colTotal = 0;
	double flopsPerPass = 0;
	for (int p = 0; p < count; p++) {
		if (ns[p] <= 0 || nd[p] <= 0) return BHIP_OK;  // empty problems go through the exact path (trivial there)
		if (srcOff[p] + ns[p] > 0x7fffffffLL || dstOff[p] + nd[p] > 0x7fffffffLL) return BHIP_OK;
		probs[p] = {(int)srcOff[p], ns[p], (int)dstOff[p], nd[p], (int)rowTotal, (int)colTotal};
		maxSrcRow = std::max(maxSrcRow, srcOff[p] + ns[p]);
		maxDstRow = std::max(maxDstRow, dstOff[p] + nd[p]);
		rowTotal += ns[p];
		colTotal += nd[p];
		flopsPerPass += 2.0 * ns[p] * (double)nd[p] * 64;
		if (rowTotal > 0x3fffffffLL || colTotal > 0x3fffffffLL) return BHIP_OK;
	}
	// block table: one block per (problem, column strip of <= STRIP_COLS columns, row chunk); the rows are split only when there are too few
	// strips to fill the chip (a block then accumulates column minima over its chunk and the chunks meet in the global atomics)
	long long strips = 0;
	for (int p = 0; p < count; p++) strips += (nd[p] + STRIP_COLS - 1) / STRIP_COLS;
	int rowSplit = (int)std::max<long long>(1, (1024 + strips - 1) / strips);
	{ const char* e = getenv("BHIP_ASSOC_ROWSPLIT"); if (e && atoi(e) > 0) rowSplit = atoi(e); }   // tests: force row chunks / whole-problem sweeps
	for (int p = 0; p < count; p++) {
		int chunk = (ns[p] + rowSplit - 1) / rowSplit;
		chunk = std::max(256, ((chunk + 255) / 256) * 256);   // whole turns of the four waves
		for (int c0 = 0; c0 < nd[p]; c0 += STRIP_COLS)
			for (int r0 = 0; r0 < ns[p]; r0 += chunk) blocks.push_back({p, r0, std::min(ns[p], r0 + chunk), c0, std::min(nd[p], c0 + STRIP_COLS)});
	}
	if (count >= 16) {
		// Workgroup ids go round-robin over the 8 XCDs: give all blocks of a problem the same id residue so that its destination rows are
		// fetched into ONE XCD's L2 instead of eight (problems are dealt to the residues in turn; short queues are padded with empty entries).
		std::vector<AssocBlock> q[8];
		for (const AssocBlock& b : blocks) q[b.p & 7].push_back(b);
		size_t longest = 0;
		for (int x = 0; x < 8; x++) longest = std::max(longest, q[x].size());
		blocks.assign(longest * 8, AssocBlock{0, 0, 0, 0, 0});
		for (int x = 0; x < 8; x++)
			for (size_t k = 0; k < q[x].size(); k++) blocks[k * 8 + x] = q[x][k];
	}
	const bool shared = dev_src == dev_dst;
	const long long rowsS = shared ? std::max(maxSrcRow, maxDstRow) : maxSrcRow;
	const int cap = (int)std::min<long long>(0x3fffffffLL, 8 * (rowTotal + colTotal) + 4096);

	BHIP_TRY(W.Fs.reserve(ctx, (size_t)rowsS * ROW_CHUNKS * 16));
	BHIP_TRY(W.nrmS.reserve(ctx, (size_t)rowsS * 4));
	if (!shared) {
		BHIP_TRY(W.Fd.reserve(ctx, (size_t)maxDstRow * ROW_CHUNKS * 16));
		BHIP_TRY(W.nrmD.reserve(ctx, (size_t)maxDstRow * 4));
	}
	BHIP_TRY(W.probs.reserve(ctx, probs.size() * sizeof(AssocProblem)));
	BHIP_TRY(W.blocks.reserve(ctx, blocks.size() * sizeof(AssocBlock)));
	BHIP_TRY(W.keys.reserve(ctx, (size_t)(rowTotal + colTotal) * 4));
	BHIP_TRY(W.thr.reserve(ctx, (size_t)(rowTotal + colTotal) * 4));
	BHIP_TRY(W.best.reserve(ctx, (size_t)(rowTotal + colTotal) * 8));
	BHIP_TRY(W.args.reserve(ctx, (size_t)(rowTotal + 2 * colTotal) * 4));
	BHIP_TRY(W.cand.reserve(ctx, (size_t)cap * sizeof(AssocCand)));
	BHIP_TRY(W.flags.reserve(ctx, 64));

	hipStream_t st = ctx->stream;
	BHIP_HIP(ctx, hipMemcpyAsync(W.probs.p, probs.data(), probs.size() * sizeof(AssocProblem), hipMemcpyHostToDevice, st));
	BHIP_HIP(ctx, hipMemcpyAsync(W.blocks.p, blocks.data(), blocks.size() * sizeof(AssocBlock), hipMemcpyHostToDevice, st));
	BHIP_HIP(ctx, hipMemsetAsync(W.flags.p, 0, 64, st));
	int* flags = W.flags.as<int>();       // [0] non-finite, [1] max norm bits, [2] number of listed pairs
	int* counters = flags + 2;
	_Float16* Hs = W.Fs.as<_Float16>();
	float* nS = W.nrmS.as<float>();
	_Float16* Hd = shared ? Hs : W.Fd.as<_Float16>();
	float* nD = shared ? nS : W.nrmD.as<float>();
	{
		ProfScope ps(ctx, "k_assoc_prep", (double)(rowsS + (shared ? 0 : maxDstRow)) * (2 * 64 * 8 + ROW_CHUNKS * 16));
		const unsigned gS = (unsigned)((rowsS + 16 * PREP_ROWS - 1) / (16 * PREP_ROWS)), gD = (unsigned)((maxDstRow + 16 * PREP_ROWS - 1) / (16 * PREP_ROWS));
		hipLaunchKernelGGL(k_assoc_norms, dim3(gS), dim3(256), 0, st, dev_src, rowsS, nS, flags);
		if (!shared) hipLaunchKernelGGL(k_assoc_norms, dim3(gD), dim3(256), 0, st, dev_dst, maxDstRow, nD, flags);
		hipLaunchKernelGGL(k_assoc_half, dim3(gS), dim3(256), 0, st, dev_src, rowsS, nS, flags, Hs);
		if (!shared) hipLaunchKernelGGL(k_assoc_half, dim3(gD), dim3(256), 0, st, dev_dst, maxDstRow, nD, flags, Hd);
	}
	unsigned int* rowKey = W.keys.as<unsigned int>();
	unsigned int* colKey = rowKey + rowTotal;
	float* rowThr = W.thr.as<float>();
	float* colThr = rowThr + rowTotal;
	unsigned long long* rowBest = W.best.as<unsigned long long>();
	unsigned long long* colBest = rowBest + rowTotal;
	int* rowArg = W.args.as<int>();
	int* colArg = rowArg + rowTotal;
	int* colCnt = colArg + colTotal;
	const long long nk = rowTotal + colTotal;
	hipLaunchKernelGGL(k_fill_u32, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, st, rowKey, nk, 0xFFFFFFFFu);
	BHIP_HIP(ctx, hipMemsetAsync(W.best.p, 0xFF, (size_t)nk * 8, st));                       // +"infinity" for the u64 minima
	BHIP_HIP(ctx, hipMemsetAsync(rowArg, 0xFF, (size_t)(rowTotal + colTotal) * 4, st));      // -1
	BHIP_HIP(ctx, hipMemsetAsync(colCnt, 0, (size_t)colTotal * 4, st));

	MfmaArgs A;
	A.Hs = (const uint4*)Hs; A.Hd = (const uint4*)Hd; A.probs = W.probs.as<AssocProblem>(); A.blocks = W.blocks.as<AssocBlock>();
	A.rowKey = rowKey; A.colKey = colKey; A.rowThr = rowThr; A.colThr = colThr;
	A.cand = W.cand.as<AssocCand>(); A.counter = counters; A.cap = cap;
#ifdef BHIP_EXPERIMENTS
	{ const char* e = getenv("BHIP_ASSOC_ABLATE"); A.ablate = e ? atoi(e) : 0; }
#endif
	const unsigned nblocks = (unsigned)blocks.size();
	{
		ProfScope ps(ctx, "k_assoc_mfma_pass1", 0, flopsPerPass);
		hipLaunchKernelGGL(k_assoc_mfma<1>, dim3(nblocks), dim3(256), 0, st, A);
	}
	int maxN = 0;
	for (int p = 0; p < count; p++) maxN = std::max(maxN, ns[p] + nd[p]);
	hipLaunchKernelGGL(k_assoc_thresholds, dim3((maxN + 255) / 256, count), dim3(256), 0, st, A.probs, count, nS, nD, rowKey, colKey, flags, rowThr, colThr);
	{
		ProfScope ps(ctx, "k_assoc_mfma_pass2", 0, flopsPerPass);
		hipLaunchKernelGGL(k_assoc_mfma<2>, dim3(nblocks), dim3(256), 0, st, A);
	}
	{
		ProfScope ps(ctx, "k_assoc_exact");
		// about one listed pair per thread (the list length is only known on the device: 2.3 per row is typical, the grid-stride loop takes the rest)
		const unsigned exactBlocks = (unsigned)std::min<long long>(16384, std::max<long long>(256, (3 * (rowTotal + colTotal) / 2 + 255) / 256));
		hipLaunchKernelGGL(k_assoc_exact, dim3(exactBlocks), dim3(256), 0, st, dev_src, dev_dst, A.probs, A.cand, counters, cap, rowBest, colBest);
		hipLaunchKernelGGL(k_assoc_argsel, dim3(exactBlocks), dim3(256), 0, st, A.probs, A.cand, counters, cap, rowBest, colBest, rowArg, colArg, colCnt);
	}
	// degenerate inputs? (one small read-back; the result kernels below are only trusted when the flags are clean)
	BHIP_HIP(ctx, hipMemcpyAsync(ctx->hostScratch, flags, 16, hipMemcpyDeviceToHost, st));
	BHIP_HIP(ctx, hipStreamSynchronize(st));
	const int* hf = ctx->hostScratch;
	if (hf[0] != 0 || hf[2] > cap) return BHIP_OK;  // *usedMfma stays 0
	int maxNs = 0;
	for (int p = 0; p < count; p++) maxNs = std::max(maxNs, ns[p]);
	hipLaunchKernelGGL(k_assoc_resolve, dim3((maxNs + 255) / 256, count), dim3(256), 0, st, A.probs, maxErr, backwards, rowBest, rowArg, colArg, colCnt,
					   dev_pairs, dev_fit);
	BHIP_HIP(ctx, hipGetLastError());
	*usedMfma = 1;
	return BHIP_OK;
}

void bhip_assoc_mfma_release(AssocMfmaWork& W) {
	DevBuf* b[] = {&W.Fs, &W.Fd, &W.nrmS, &W.nrmD, &W.probs, &W.blocks, &W.keys, &W.thr, &W.best, &W.args, &W.cand, &W.flags};
	for (DevBuf* x : b) x->release();
}
