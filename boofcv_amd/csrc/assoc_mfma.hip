// K6 (MFMA path): batched greedy L2 association whose results are bit-identical to the reference's fp64 sequential scores.
//
// Reference: AssociateGreedy.associate  F:alg/feature/associate/AssociateGreedy.java:65-118 with
//            DescriptorDistance.euclideanSq  F:alg/descriptor/DescriptorDistance.java:55-64  (SURVEY hard part 3)
//
// The N x M x 64 contraction runs on the matrix cores in fp32 (v_mfma_f32_32x32x2_f32, exact f32 fma chain):
//     d~(i,j) = |a_i|^2 + |b_j|^2 - 2 <fl32(a_i), fl32(b_j)>
// with the rigorous bound |d~ - d| <= eps(i,j) = 71 u (|a_i|^2 + |b_j|^2), u = 2^-24
//   (input rounding 2u|a||b|, 64-term fma chain 64u|a||b|, both doubled by the factor 2, |a||b| <= (|a|^2+|b|^2)/2, epilogue <= 4u(|a|^2+|b|^2)).
// Pass 1 reduces d~ to per-row and per-column minima.  Pass 2 recomputes the tiles and lists every pair inside the band
//     d~(i,j) <= rowmin~(i) + band(i)   /   d~(i,j) <= colmin~(j) + band(j),     band = 2 * 160 u (|.|^2 + max|.|^2)  (> 2 eps)
// The true row arg-min (and every exact tie of it) and the true column minimum (and every exact tie) are provably inside those
// lists, and everything outside is strictly larger than the list's best.  The listed pairs (about one per row/column) are then
// re-scored with the exact sequential fp64 loop and the reference's rules are applied to the exact values:
//   forward : smallest exact score, largest destination index among exact ties, inclusive maxFitError
//   backward: (i -> m) survives iff i is the only row attaining the exact minimum of column m
// Degenerate inputs (candidate list overflow, e.g. all descriptors equal; non-finite values) fall back to the exact VALU kernels.
//
// Layout: descriptors are converted once per call to fp32 rows [row][64]; lane l of a wave holds k in [32(l>>5), 32(l>>5)+32) of row/col
// (l&31), which is exactly the A / B fragment order of 32 consecutive 32x32x2 MFMAs -- operands stay in VGPRs, no LDS.
// Each wave owns 32 source rows and sweeps destination columns 64 at a time (two accumulators).
// Bound: MFMA (fp32 matrix rate, 157 TFLOP/s); algorithmic flops per pass = 2 * Ns * Nd * 64.
#include "common.h"
#include <cfloat>
#include <algorithm>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct AssocProblem {
	int srcOff, ns, dstOff, nd;   // rows in the src / dst descriptor buffers
	int rowBase, colBase;         // compact offsets into the per-row / per-column work arrays
};
struct AssocBlock {
	int p, row0, col0, col1;
};
struct AssocCand {
	int p, i, j, pad;
	double score;
};

#define BAND_C 160.0f
#define U24 5.9604644775390625e-08f

__device__ __forceinline__ unsigned int fkey(float f) {
	const unsigned int b = __float_as_uint(f);
	return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float fkeyInv(unsigned int k) {
	return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

// ---- fp64 -> fp32 rows + squared norms; flags[0] = non-finite seen, flags[1] = max norm (float bits, non-negative) ----
__global__ __launch_bounds__(256) void k_assoc_prep(const double* __restrict__ D, long long rows, float* __restrict__ F, float* __restrict__ nrm,
													  int* __restrict__ flags) {
	const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	if (row >= rows) return;
	const double v = D[row * 64 + lane];
	F[row * 64 + lane] = (float)v;
	double s = v * v;
#pragma unroll
	for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
	if (lane == 0) {
		const float n = (float)s;
		// round the norm up so the band computed from it can only grow
		const float nUp = n * (1.0f + 4.0f * U24);
		nrm[row] = nUp;
		// one hot word for the whole launch: only touch it when this row actually raises the maximum (a handful of times per launch)
		if (!(s < 1e30)) atomicOr(&flags[0], 1);  // NaN, Inf or absurdly large: use the exact path
		else if (__float_as_int(nUp) > __hip_atomic_load(&flags[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&flags[1], __float_as_int(nUp));
	}
}

__global__ void k_fill_u32(unsigned int* p, long long n, unsigned int v) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) p[i] = v;
}

// thr = min~ + band, band = 2 * 160 u (norm + maxNorm)
__global__ void k_assoc_thresholds(const AssocProblem* __restrict__ probs, int count, const float* __restrict__ nrmS, const float* __restrict__ nrmD,
								   const unsigned int* __restrict__ rowKey, const unsigned int* __restrict__ colKey, const int* __restrict__ flags,
								   float* __restrict__ rowThr, float* __restrict__ colThr) {
	const int p = blockIdx.y;
	const AssocProblem P = probs[p];
	const float maxN = __int_as_float(flags[1]);
	for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < P.ns + P.nd; i += gridDim.x * blockDim.x) {
		if (i < P.ns) {
			const float band = 2.0f * BAND_C * U24 * (nrmS[P.srcOff + i] + maxN);
			rowThr[P.rowBase + i] = fkeyInv(rowKey[P.rowBase + i]) + band;
		} else {
			const int j = i - P.ns;
			const float band = 2.0f * BAND_C * U24 * (nrmD[P.dstOff + j] + maxN);
			colThr[P.colBase + j] = fkeyInv(colKey[P.colBase + j]) + band;
		}
	}
}

struct MfmaArgs {
	const float* Fs;
	const float* Fd;
	const float* nrmS;
	const float* nrmD;
	const AssocProblem* probs;
	const AssocBlock* blocks;
	unsigned int* rowKey;
	unsigned int* colKey;
	const float* rowThr;
	const float* colThr;
	AssocCand* rowCand;
	AssocCand* colCand;
	int* counters;   // [0] row candidates, [1] column candidates
	int cap;
};

#define CAND_BUF 96   // per-wave, per-list staging slots in LDS (>= 64: one ballot can add up to 64 entries)

// Wave-aggregated candidate lists: hits are staged in LDS and flushed with ONE atomicAdd per flush, so the two global counters see a
// few atomics per wave instead of one per candidate (a single contended word sustains only ~88 atomics/us, MI355X_MICROARCH.md).
struct CandStage {
	int4* buf;       // this wave's [2][CAND_BUF]
	int cnt[2];      // wave-uniform
};
__device__ __forceinline__ void candFlush(const MfmaArgs& A, CandStage& S, int list, int lane) {
	const int n = S.cnt[list];
	if (n == 0) return;
	int base = 0;
	if (lane == 0) base = atomicAdd(&A.counters[list], n);
	base = __builtin_amdgcn_readfirstlane(base);
	AssocCand* out = list == 0 ? A.rowCand : A.colCand;
	for (int k = lane; k < n; k += 64) {
		if (base + k < A.cap) {
			const int4 v = S.buf[list * CAND_BUF + k];
			AssocCand c; c.p = v.x; c.i = v.y; c.j = v.z; c.pad = 0; c.score = 0;
			out[base + k] = c;
		}
	}
	S.cnt[list] = 0;
}
__device__ __forceinline__ void candPush(const MfmaArgs& A, CandStage& S, int list, bool cond, int p, int i, int j, int lane) {
	const unsigned long long m = __ballot(cond);
	if (m == 0) return;
	const int n = __popcll(m);
	if (S.cnt[list] + n > CAND_BUF) candFlush(A, S, list, lane);
	if (cond) {
		const int pos = S.cnt[list] + __popcll(m & ((1ull << lane) - 1ull));
		S.buf[list * CAND_BUF + pos] = make_int4(p, i, j, 0);
	}
	S.cnt[list] += n;
}

template <int PASS>
__global__ __launch_bounds__(256, 2) void k_assoc_mfma(MfmaArgs A) {
	__shared__ int4 candLds[4][2 * CAND_BUF];
	const AssocBlock B = A.blocks[blockIdx.x];
	const AssocProblem P = A.probs[B.p];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int r = lane & 31, h = lane >> 5;
	const int rowT = B.row0 + 32 * wave;  // first row of this wave's tile
	const bool live = rowT < P.ns;        // a wave without rows still stages B tiles and meets the barriers
	CandStage S;
	S.buf = candLds[wave];
	S.cnt[0] = 0;
	S.cnt[1] = 0;

	// A fragment: row rowT + r, k in [32h, 32h+32)
	float a[32];
	{
		const int row = rowT + r;
		if (row < P.ns) {
			const f32x4* src = (const f32x4*)(A.Fs + ((long long)(P.srcOff + row) * 64 + 32 * h));
#pragma unroll
			for (int q = 0; q < 8; q++) {
				const f32x4 v = src[q];
				a[4 * q] = v.x; a[4 * q + 1] = v.y; a[4 * q + 2] = v.z; a[4 * q + 3] = v.w;
			}
		} else {
#pragma unroll
			for (int q = 0; q < 32; q++) a[q] = 0.0f;
		}
	}
	// per accumulator register: the row it belongs to (C layout: row = (reg&3) + 8*(reg>>2) + 4h, col = lane&31)
	float rowN[16];   // |a|^2 of that row (INF when the row does not exist)
	float rowV[16];   // PASS 1: running row minimum ; PASS 2: row threshold
#pragma unroll
	for (int g = 0; g < 16; g++) {
		const int rr = rowT + (g & 3) + 8 * (g >> 2) + 4 * h;
		const bool ok = rr < P.ns;
		rowN[g] = ok ? A.nrmS[P.srcOff + rr] : INFINITY;
		if (PASS == 1) rowV[g] = INFINITY;
		else rowV[g] = ok ? A.rowThr[P.rowBase + rr] : -INFINITY;
	}

	// The four waves of a block sweep the same destination columns, so every 64-column step of B is staged ONCE per block in LDS
	// (fp32 rows as 16-byte chunks, [tile][chunk][column] with a pitch of 33 chunks: fragment reads are contiguous across lanes) and
	// double-buffered: the next step's chunks are fetched into registers before this step's MFMAs and stored after them.  One L2 read
	// of B per block instead of four, and its latency sits behind the matrix work.
	__shared__ uint4 tileB[2][2][16 * 33];
	const int tid = threadIdx.x;
	uint4 pre[4];
	auto fetch = [&](int c0) {
#pragma unroll
		for (int q = 0; q < 4; q++) {
			const int g = tid + 256 * q;          // 2 tiles x 32 columns x 16 chunks
			const int col = c0 + (g >> 4);        // (g >> 9) * 32 + ((g & 511) >> 4) == g >> 4
			pre[q] = col < B.col1 ? ((const uint4*)(A.Fd + (long long)(P.dstOff + col) * 64))[g & 15] : make_uint4(0, 0, 0, 0);
		}
	};
	auto stash = [&](int buf) {
#pragma unroll
		for (int q = 0; q < 4; q++) {
			const int g = tid + 256 * q;
			tileB[buf][g >> 9][(g & 15) * 33 + ((g & 511) >> 4)] = pre[q];
		}
	};
	fetch(B.col0);
	stash(0);
	__syncthreads();
	int buf = 0;
	for (int c0 = B.col0; c0 < B.col1; c0 += 64, buf ^= 1) {
		const int colA = c0 + r, colB = c0 + 32 + r;
		const bool okA = colA < B.col1, okB = colB < B.col1;
		const bool more = c0 + 64 < B.col1;   // block-uniform
		if (more) fetch(c0 + 64);
		const float nbA = okA ? A.nrmD[P.dstOff + colA] : INFINITY;
		const float nbB = okB ? A.nrmD[P.dstOff + colB] : INFINITY;
		f32x16 acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
		f32x16 acc1 = acc0;
		// lane (r, h) needs k in [32h, 32h+32) of columns colA / colB = chunks 8h .. 8h+7; two halves keep 32 B registers live, not 64
#pragma unroll
		for (int half = 0; half < 2; half++) {
			float b0[16], b1[16];
#pragma unroll
			for (int q = 0; q < 4; q++) {
				const uint4 v = tileB[buf][0][(8 * h + 4 * half + q) * 33 + r], w = tileB[buf][1][(8 * h + 4 * half + q) * 33 + r];
				b0[4 * q] = __uint_as_float(v.x); b0[4 * q + 1] = __uint_as_float(v.y); b0[4 * q + 2] = __uint_as_float(v.z); b0[4 * q + 3] = __uint_as_float(v.w);
				b1[4 * q] = __uint_as_float(w.x); b1[4 * q + 1] = __uint_as_float(w.y); b1[4 * q + 2] = __uint_as_float(w.z); b1[4 * q + 3] = __uint_as_float(w.w);
			}
#pragma unroll
			for (int s = 0; s < 16; s++) {
				acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[16 * half + s], b0[s], acc0, 0, 0, 0);
				acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[16 * half + s], b1[s], acc1, 0, 0, 0);
			}
		}
		if (more) stash(buf ^ 1);   // the other buffer was last read before the previous barrier
		if (live) {
		if (PASS == 1) {
			float cA = INFINITY, cB = INFINITY;
#pragma unroll
			for (int g = 0; g < 16; g++) {
				const float dA = (rowN[g] + nbA) - 2.0f * acc0[g];
				const float dB = (rowN[g] + nbB) - 2.0f * acc1[g];
				rowV[g] = fminf(rowV[g], fminf(dA, dB));
				cA = fminf(cA, dA);
				cB = fminf(cB, dB);
			}
			cA = fminf(cA, __shfl_xor(cA, 32, 64));
			cB = fminf(cB, __shfl_xor(cB, 32, 64));
			if (h == 0) {
				if (okA) atomicMin(&A.colKey[P.colBase + colA], fkey(cA));
				if (okB) atomicMin(&A.colKey[P.colBase + colB], fkey(cB));
			}
		} else {
			const float tA = okA ? A.colThr[P.colBase + colA] : -INFINITY;
			const float tB = okB ? A.colThr[P.colBase + colB] : -INFINITY;
			// hit bits first (branch-free), then the rare staging work only for registers that have a hit somewhere in the wave
			unsigned int hits = 0;
#pragma unroll
			for (int g = 0; g < 16; g++) {
				const float dA = (rowN[g] + nbA) - 2.0f * acc0[g];
				const float dB = (rowN[g] + nbB) - 2.0f * acc1[g];
				hits |= (dA <= rowV[g] || dB <= rowV[g] || dA <= tA || dB <= tB) ? (1u << g) : 0u;
			}
			unsigned int any = hits;
#pragma unroll
			for (int o = 32; o >= 1; o >>= 1) any |= __shfl_xor(any, o, 64);
			any = __builtin_amdgcn_readfirstlane(any);
#pragma unroll
			for (int g = 0; g < 16; g++) {
				if (any & (1u << g)) {
					const int rr = rowT + (g & 3) + 8 * (g >> 2) + 4 * h;
					const float dA = (rowN[g] + nbA) - 2.0f * acc0[g];   // same expression as above: identical bits
					const float dB = (rowN[g] + nbB) - 2.0f * acc1[g];
					candPush(A, S, 0, dA <= rowV[g], B.p, rr, colA, lane);
					candPush(A, S, 0, dB <= rowV[g], B.p, rr, colB, lane);
					candPush(A, S, 1, dA <= tA, B.p, rr, colA, lane);
					candPush(A, S, 1, dB <= tB, B.p, rr, colB, lane);
				}
			}
		}
		}   // live
		__syncthreads();
	}
	if (!live) return;
	if (PASS == 2) {
		candFlush(A, S, 0, lane);
		candFlush(A, S, 1, lane);
	}
	if (PASS == 1) {
#pragma unroll
		for (int g = 0; g < 16; g++) {
			float v = rowV[g];
#pragma unroll
			for (int o = 16; o >= 1; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
			const int rr = rowT + (g & 3) + 8 * (g >> 2) + 4 * h;
			if (r == 0 && rr < P.ns) atomicMin(&A.rowKey[P.rowBase + rr], fkey(v));
		}
	}
}

// exact sequential fp64 score of every listed pair + exact minima per row / column (scores are >= 0: bit order == value order)
__global__ __launch_bounds__(256) void k_assoc_exact(const double* __restrict__ src, const double* __restrict__ dst, const AssocProblem* __restrict__ probs,
													   AssocCand* __restrict__ rowCand, AssocCand* __restrict__ colCand, const int* __restrict__ counters, int cap,
													   unsigned long long* __restrict__ rowBest, unsigned long long* __restrict__ colBest) {
	const int nRow = min(counters[0], cap), nCol = min(counters[1], cap);
	for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < nRow + nCol; t += gridDim.x * blockDim.x) {
		const bool isRow = t < nRow;
		AssocCand* c = isRow ? &rowCand[t] : &colCand[t - nRow];
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
		if (isRow) atomicMin(&rowBest[P.rowBase + c->i], bits);
		else atomicMin(&colBest[P.colBase + c->j], bits);
	}
}

// among the pairs that attain the exact minimum: largest destination index per row; count + any source index per column
__global__ __launch_bounds__(256) void k_assoc_argsel(const AssocProblem* __restrict__ probs, const AssocCand* __restrict__ rowCand,
														const AssocCand* __restrict__ colCand, const int* __restrict__ counters, int cap,
														const unsigned long long* __restrict__ rowBest, const unsigned long long* __restrict__ colBest,
														int* __restrict__ rowArg, int* __restrict__ colArg, int* __restrict__ colCnt) {
	const int nRow = min(counters[0], cap), nCol = min(counters[1], cap);
	for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < nRow + nCol; t += gridDim.x * blockDim.x) {
		const bool isRow = t < nRow;
		const AssocCand c = isRow ? rowCand[t] : colCand[t - nRow];
		const AssocProblem P = probs[c.p];
		const unsigned long long bits = (unsigned long long)__double_as_longlong(c.score);
		if (isRow) {
			if (bits == rowBest[P.rowBase + c.i]) atomicMax(&rowArg[P.rowBase + c.i], c.j);
		} else {
			if (bits == colBest[P.colBase + c.j]) {
				atomicAdd(&colCnt[P.colBase + c.j], 1);
				atomicMax(&colArg[P.colBase + c.j], c.i);
			}
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
	// block table: 128 rows per block; split the columns when there are too few row blocks to fill the chip
	long long rowBlocks = 0;
	for (int p = 0; p < count; p++) rowBlocks += (ns[p] + 127) / 128;
	int colSplit = (int)std::max<long long>(1, (1024 + rowBlocks - 1) / rowBlocks);
	{ const char* e = getenv("BHIP_ASSOC_COLSPLIT"); if (e && atoi(e) > 0) colSplit = atoi(e); }   // tests: force long column sweeps per block
	for (int p = 0; p < count; p++) {
		int split = std::min(colSplit, (nd[p] + 63) / 64);
		int per = (nd[p] + split - 1) / split;
		per = ((per + 63) / 64) * 64;
		for (int r0 = 0; r0 < ns[p]; r0 += 128)
			for (int c0 = 0; c0 < nd[p]; c0 += per) blocks.push_back({p, r0, c0, std::min(nd[p], c0 + per)});
	}
	const bool shared = dev_src == dev_dst;
	const long long rowsS = shared ? std::max(maxSrcRow, maxDstRow) : maxSrcRow;
	const int cap = (int)std::min<long long>(0x3fffffffLL, 4 * (rowTotal + colTotal) + 4096);

	BHIP_TRY(W.Fs.reserve(ctx, (size_t)rowsS * 64 * 4));
	BHIP_TRY(W.nrmS.reserve(ctx, (size_t)rowsS * 4));
	if (!shared) {
		BHIP_TRY(W.Fd.reserve(ctx, (size_t)maxDstRow * 64 * 4));
		BHIP_TRY(W.nrmD.reserve(ctx, (size_t)maxDstRow * 4));
	}
	BHIP_TRY(W.probs.reserve(ctx, probs.size() * sizeof(AssocProblem)));
	BHIP_TRY(W.blocks.reserve(ctx, blocks.size() * sizeof(AssocBlock)));
	BHIP_TRY(W.keys.reserve(ctx, (size_t)(rowTotal + colTotal) * 4));
	BHIP_TRY(W.thr.reserve(ctx, (size_t)(rowTotal + colTotal) * 4));
	BHIP_TRY(W.best.reserve(ctx, (size_t)(rowTotal + colTotal) * 8));
	BHIP_TRY(W.args.reserve(ctx, (size_t)(rowTotal + 2 * colTotal) * 4));
	BHIP_TRY(W.cand.reserve(ctx, (size_t)cap * 2 * sizeof(AssocCand)));
	BHIP_TRY(W.flags.reserve(ctx, 64));

	hipStream_t st = ctx->stream;
	BHIP_HIP(ctx, hipMemcpyAsync(W.probs.p, probs.data(), probs.size() * sizeof(AssocProblem), hipMemcpyHostToDevice, st));
	BHIP_HIP(ctx, hipMemcpyAsync(W.blocks.p, blocks.data(), blocks.size() * sizeof(AssocBlock), hipMemcpyHostToDevice, st));
	BHIP_HIP(ctx, hipMemsetAsync(W.flags.p, 0, 64, st));
	int* flags = W.flags.as<int>();       // [0] non-finite, [1] max norm bits, [2],[3] candidate counters
	int* counters = flags + 2;
	float* Fs = W.Fs.as<float>();
	float* nS = W.nrmS.as<float>();
	float* Fd = shared ? Fs : W.Fd.as<float>();
	float* nD = shared ? nS : W.nrmD.as<float>();
	{
		ProfScope ps(ctx, "k_assoc_prep", (double)rowsS * 64 * 12);
		hipLaunchKernelGGL(k_assoc_prep, dim3((unsigned)((rowsS + 3) / 4)), dim3(256), 0, st, dev_src, rowsS, Fs, nS, flags);
		if (!shared) hipLaunchKernelGGL(k_assoc_prep, dim3((unsigned)((maxDstRow + 3) / 4)), dim3(256), 0, st, dev_dst, maxDstRow, Fd, nD, flags);
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
	A.Fs = Fs; A.Fd = Fd; A.nrmS = nS; A.nrmD = nD; A.probs = W.probs.as<AssocProblem>(); A.blocks = W.blocks.as<AssocBlock>();
	A.rowKey = rowKey; A.colKey = colKey; A.rowThr = rowThr; A.colThr = colThr;
	A.rowCand = W.cand.as<AssocCand>(); A.colCand = A.rowCand + cap; A.counters = counters; A.cap = cap;
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
		hipLaunchKernelGGL(k_assoc_exact, dim3(1024), dim3(256), 0, st, dev_src, dev_dst, A.probs, A.rowCand, A.colCand, counters, cap, rowBest, colBest);
		hipLaunchKernelGGL(k_assoc_argsel, dim3(1024), dim3(256), 0, st, A.probs, A.rowCand, A.colCand, counters, cap, rowBest, colBest, rowArg, colArg, colCnt);
	}
	// degenerate inputs? (one small read-back; the result kernels below are only trusted when the flags are clean)
	BHIP_HIP(ctx, hipMemcpyAsync(ctx->hostScratch, flags, 16, hipMemcpyDeviceToHost, st));
	BHIP_HIP(ctx, hipStreamSynchronize(st));
	const int* hf = ctx->hostScratch;
	if (hf[0] != 0 || hf[2] > cap || hf[3] > cap) return BHIP_OK;  // *usedMfma stays 0
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
