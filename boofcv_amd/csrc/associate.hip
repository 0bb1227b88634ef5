// K6 / K7: greedy brute-force association with exact scores (fp64 sequential L2, integer Hamming).
//
// Reference:
//   AssociateGreedy.associate          F:alg/feature/associate/AssociateGreedy.java:65-118
//   DescriptorDistance.euclideanSq     F:alg/descriptor/DescriptorDistance.java:55-64   (euclidean :36-46)
//   DescriptorDistance.hamming         F:alg/descriptor/DescriptorDistance.java:196-220
//
// The reference stores the whole Ns x Nd score matrix and then walks its columns.  Here the matrix is never stored:
//   forward : per source row, arg-min over destinations with `fit <= best` (largest index among minima, inclusive threshold)
//   backward: per destination column, top-2 over sources (min1, argmin1, min2).  Match (i -> m) survives iff i is the arg-min of
//             column m and min2 > min1 -- exactly "no other j has W[j][m] <= W[i][m]" (AssociateGreedy.java:105-114).
// Both passes evaluate the same sequential expression per pair, so scores are bit-identical to the reference's and to each other.
// Each thread keeps one descriptor in registers and streams the other set through LDS (broadcast reads).  The column pass of a sharded
// problem writes (min1, min2, argmin1) records that are all-gathered across ranks and merged by k_assoc_finish.
#include "common.h"
#include <cfloat>
#include <cmath>
#include <algorithm>

struct ColTop {
	double min1, min2;
	int idx1, pad;
};
struct RowBest {
	double best;
	int idx, pad;
};

#define VT 64  // vectors of the streamed set per LDS tile

// int8 MFMA Hamming path (assoc_ham_mfma.hip)
int bhip_ham_expand(bhip_ctx* ctx, const int* D, long long rows, int words, unsigned char* bytes, int* pop);
int bhip_ham_mfma_splits(int nU, int nV);
int bhip_ham_mfma_scan(bhip_ctx* ctx, bool colMode, const unsigned char* Ub, const int* Up, int nU, const unsigned char* Vb, const int* Vp, int nV, int words,
						 int vBase, double maxErr, void* partial, int splits);

template <int DOF>
struct L2Scorer {
	typedef double elem;
	double a[DOF];
	int sqrtScore;
	__device__ __forceinline__ void load(const double* __restrict__ u, int) {
#pragma unroll
		for (int k = 0; k < DOF; k++) a[k] = u[k];
	}
	__device__ __forceinline__ double score(const double* __restrict__ v, int) const {
		double total = 0;
#pragma unroll
		for (int k = 0; k < DOF; k++) {
			const double d = a[k] - v[k];
			total += d * d;
		}
		return sqrtScore ? sqrt(total) : total;
	}
};
// run-time length: the owned descriptor is re-read from global memory (L1/L2 resident)
struct L2ScorerDyn {
	typedef double elem;
	const double* u;
	int sqrtScore;
	__device__ __forceinline__ void load(const double* __restrict__ p, int) { u = p; }
	__device__ __forceinline__ double score(const double* __restrict__ v, int n) const {
		double total = 0;
		for (int k = 0; k < n; k++) {
			const double d = u[k] - v[k];
			total += d * d;
		}
		return sqrtScore ? sqrt(total) : total;
	}
};
template <int WORDS>
struct HamScorer {
	typedef int elem;
	int a[WORDS];
	int sqrtScore;
	__device__ __forceinline__ void load(const int* __restrict__ u, int) {
#pragma unroll
		for (int k = 0; k < WORDS; k++) a[k] = u[k];
	}
	__device__ __forceinline__ double score(const int* __restrict__ v, int) const {
		int s = 0;
#pragma unroll
		for (int k = 0; k < WORDS; k++) s += __popc((unsigned)(a[k] ^ v[k]));
		return (double)s;
	}
};
struct HamScorerDyn {
	typedef int elem;
	const int* u;
	int sqrtScore;
	__device__ __forceinline__ void load(const int* __restrict__ p, int) { u = p; }
	__device__ __forceinline__ double score(const int* __restrict__ v, int n) const {
		int s = 0;
		for (int k = 0; k < n; k++) s += __popc((unsigned)(u[k] ^ v[k]));
		return (double)s;
	}
};

// One thread owns vector `ui` of set U and scans vectors [v0, v1) of set V.
//   COLMODE = false: U = sources,      V = destinations -> RowBest partial  (out[split][ui])
//   COLMODE = true : U = destinations, V = sources      -> ColTop  partial  (out[split][ui]), idx = vBase + v
template <class Scorer, bool COLMODE>
__global__ __launch_bounds__(256) void k_assoc_scan(const typename Scorer::elem* __restrict__ U, int nU, const typename Scorer::elem* __restrict__ V, int nV,
													 int len, int vPerSplit, int vBase, double maxErr, int sqrtScore, void* __restrict__ outRaw) {
	typedef typename Scorer::elem elem;
	extern __shared__ __attribute__((aligned(16))) unsigned char ldsRaw[];
	elem* tile = (elem*)ldsRaw;
	const int ui = blockIdx.x * blockDim.x + threadIdx.x;
	const int split = blockIdx.y;
	const int v0 = split * vPerSplit;
	const int v1 = min(nV, v0 + vPerSplit);
	const bool active = ui < nU;
	Scorer sc;
	sc.sqrtScore = sqrtScore;
	sc.load(U + (long long)(active ? ui : 0) * len, len);

	double best = maxErr, min1 = INFINITY, min2 = INFINITY;
	int bestIdx = -1;
	for (int t0 = v0; t0 < v1; t0 += VT) {
		const int nt = min(VT, v1 - t0);
		__syncthreads();
		const long long base = (long long)t0 * len;
		for (int e = threadIdx.x; e < nt * len; e += blockDim.x) tile[e] = V[base + e];
		__syncthreads();
		if (active) {
			for (int j = 0; j < nt; j++) {
				const double fit = sc.score(tile + j * len, len);
				if (!COLMODE) {
					if (fit <= best) { best = fit; bestIdx = t0 + j; }
				} else {
					if (fit < min1) { min2 = min1; min1 = fit; bestIdx = vBase + t0 + j; }
					else if (fit < min2) { min2 = fit; }
				}
			}
		}
	}
	if (!active) return;
	if (!COLMODE) {
		RowBest* out = (RowBest*)outRaw + (long long)split * nU + ui;
		out->best = best; out->idx = bestIdx; out->pad = 0;
	} else {
		ColTop* out = (ColTop*)outRaw + (long long)split * nU + ui;
		out->min1 = min1; out->min2 = min2; out->idx1 = bestIdx; out->pad = 0;
	}
}

// merge the per-split row partials in increasing destination order (`<=` keeps the largest index among minima)
__global__ void k_merge_rows(const RowBest* __restrict__ part, int nsplit, int ns, double maxErr, int* __restrict__ pairs, double* __restrict__ fit) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= ns) return;
	double best = maxErr;
	int idx = -1;
	for (int s = 0; s < nsplit; s++) {
		const RowBest p = part[(long long)s * ns + i];
		if (p.idx >= 0 && p.best <= best) { best = p.best; idx = p.idx; }
	}
	pairs[i] = idx;
	fit[i] = best;
}

__device__ __forceinline__ void top2Merge(double& min1, double& min2, int& idx1, const ColTop& p) {
	if (p.min1 < min1) {
		min2 = fmin(min1, p.min2);   // old min1 vs the newcomer's runner-up
		min1 = p.min1;
		idx1 = p.idx1;
	} else {
		if (p.min1 < min2) min2 = p.min1;
	}
}

// merge column partials (splits of one rank, or the all-gathered records of all ranks) into one record per column
__global__ void k_merge_cols(const ColTop* __restrict__ part, int nparts, int nd, ColTop* __restrict__ out) {
	const int j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j >= nd) return;
	double min1 = INFINITY, min2 = INFINITY;
	int idx1 = -1;
	for (int s = 0; s < nparts; s++) top2Merge(min1, min2, idx1, part[(long long)s * nd + j]);
	ColTop r;
	r.min1 = min1; r.min2 = min2; r.idx1 = idx1; r.pad = 0;
	out[j] = r;
}

// backwards validation: keep (i -> m) iff i is the unique strict minimum of column m
__global__ void k_assoc_finish(const ColTop* __restrict__ col, int ncolParts, int nd, int nsLocal, int srcBegin, int* __restrict__ pairs,
							   double* __restrict__ fit) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= nsLocal) return;
	const int m = pairs[i];
	if (m < 0) return;
	double min1 = INFINITY, min2 = INFINITY;
	int idx1 = -1;
	for (int s = 0; s < ncolParts; s++) top2Merge(min1, min2, idx1, col[(long long)s * nd + m]);
	if (!(idx1 == srcBegin + i && min2 > min1)) {
		pairs[i] = -1;
		fit[i] = DBL_MAX;
	}
}

static int chooseSplits(int nU, int nV) {
	const int ublocks = (nU + 255) / 256;
	int splits = (768 + ublocks - 1) / ublocks;
	const int maxSplits = (nV + VT - 1) / VT;
	if (splits > maxSplits) splits = maxSplits;
	if (splits < 1) splits = 1;
	return splits;
}

template <class Scorer, bool COLMODE>
static int launchScan(bhip_ctx* ctx, const typename Scorer::elem* U, int nU, const typename Scorer::elem* V, int nV, int len, int vBase, double maxErr,
					  int sqrtScore, void* partial, int splits) {
	int per = (nV + splits - 1) / splits;
	per = ((per + VT - 1) / VT) * VT;
	dim3 grid((nU + 255) / 256, splits);
	const size_t lds = (size_t)VT * len * sizeof(typename Scorer::elem);
	{
		// one multiply-add pair per descriptor element per (u,v) pair: the N x M x len contraction of SURVEY 8d
		ProfScope ps(ctx, COLMODE ? "k_assoc_scan_cols" : "k_assoc_scan_rows", 0, 2.0 * nU * (double)nV * len);
		hipLaunchKernelGGL((k_assoc_scan<Scorer, COLMODE>), grid, dim3(256), lds, ctx->stream, U, nU, V, nV, len, per, vBase, maxErr, sqrtScore, partial);
	}
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}

template <bool COLMODE>
static int scanL2(bhip_ctx* ctx, const double* U, int nU, const double* V, int nV, int dof, int vBase, double maxErr, int sqrtScore, void* partial,
				  int splits) {
	if ((size_t)VT * dof * 8 > 64 * 1024) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "descriptor too long for the LDS tile");
	if (dof == 64) return launchScan<L2Scorer<64>, COLMODE>(ctx, U, nU, V, nV, dof, vBase, maxErr, sqrtScore, partial, splits);
	return launchScan<L2ScorerDyn, COLMODE>(ctx, U, nU, V, nV, dof, vBase, maxErr, sqrtScore, partial, splits);
}
template <bool COLMODE>
static int scanHam(bhip_ctx* ctx, const int* U, int nU, const int* V, int nV, int words, int vBase, double maxErr, void* partial, int splits) {
	if ((size_t)VT * words * 4 > 64 * 1024) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "descriptor too long for the LDS tile");
	if (words == 16) return launchScan<HamScorer<16>, COLMODE>(ctx, U, nU, V, nV, words, vBase, maxErr, 0, partial, splits);
	return launchScan<HamScorerDyn, COLMODE>(ctx, U, nU, V, nV, words, vBase, maxErr, 0, partial, splits);
}

static int fillUnmatched(bhip_ctx* ctx, int ns, double maxErr, int* pairs, double* fit);

__global__ void k_fill_unmatched(int ns, double maxErr, int* pairs, double* fit) {
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < ns) { pairs[i] = -1; fit[i] = maxErr; }
}
static int fillUnmatched(bhip_ctx* ctx, int ns, double maxErr, int* pairs, double* fit) {
	if (ns <= 0) return BHIP_OK;
	hipLaunchKernelGGL(k_fill_unmatched, dim3((ns + 255) / 256), dim3(256), 0, ctx->stream, ns, maxErr, pairs, fit);
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}

// phase 1 of the (possibly sharded) association; colTop receives one merged record per destination column (may be null when !needCols)
template <class E>
static int phase1(bhip_ctx* ctx, bool hamming, const E* src, int nsLocal, int srcBegin, const E* dst, int nd, int len, double maxErr, int sqrtScore,
				  int* pairs, double* fit, ColTop* colTop, DevBuf& work) {
	if (nsLocal < 0 || nd < 0 || len <= 0) return bhip_fail(ctx, BHIP_ERR_INVALID, "negative size");
	if (nd == 0 || nsLocal == 0) {
		BHIP_TRY(fillUnmatched(ctx, nsLocal, maxErr, pairs, fit));
		if (colTop && nd > 0) {
			// no local rows: every column record is empty
			const int rs = 1;
			BHIP_TRY(work.reserve(ctx, sizeof(ColTop)));
			(void)rs;
			hipLaunchKernelGGL(k_merge_cols, dim3((nd + 255) / 256), dim3(256), 0, ctx->stream, (const ColTop*)work.p, 0, nd, colTop);
			BHIP_HIP(ctx, hipGetLastError());
		}
		return BHIP_OK;
	}
	// BRIEF-512 (16 words) runs on the int8 matrix cores (assoc_ham_mfma.hip: exact integer scores, same partial records);
	// BHIP_HAM_VALU=1 keeps the popcount scan below (cross-check)
	const bool hamValu = bhip_env_flag("BHIP_HAM_VALU");   // parity cross-check: popcount scan instead of the int8 matrix cores
	const bool hamMfma = hamming && len == 16 && !hamValu;
	const int rsplits = hamMfma ? bhip_ham_mfma_splits(nsLocal, nd) : chooseSplits(nsLocal, nd);
	const int csplits = !colTop ? 0 : hamMfma ? bhip_ham_mfma_splits(nd, nsLocal) : chooseSplits(nd, nsLocal);
	const size_t rowBytes = ((size_t)rsplits * nsLocal * sizeof(RowBest) + 63) & ~(size_t)63;
	const size_t colBytes = ((size_t)csplits * nd * sizeof(ColTop) + 63) & ~(size_t)63;
	const size_t expBytes = hamMfma ? ((size_t)(nsLocal + nd) * (32 * 16 + 4) + 256) : 0;
	BHIP_TRY(work.reserve(ctx, rowBytes + colBytes + expBytes + 64));
	RowBest* rowPart = (RowBest*)work.p;
	ColTop* colPart = (ColTop*)((char*)work.p + rowBytes);
	if (hamMfma) {
		unsigned char* srcB = (unsigned char*)work.p + rowBytes + colBytes;
		unsigned char* dstB = srcB + (size_t)nsLocal * 512;
		int* srcP = (int*)(dstB + (size_t)nd * 512);
		int* dstP = srcP + nsLocal;
		BHIP_TRY(bhip_ham_expand(ctx, (const int*)src, nsLocal, 16, srcB, srcP));
		BHIP_TRY(bhip_ham_expand(ctx, (const int*)dst, nd, 16, dstB, dstP));
		BHIP_TRY(bhip_ham_mfma_scan(ctx, false, srcB, srcP, nsLocal, dstB, dstP, nd, 16, 0, maxErr, rowPart, rsplits));
		hipLaunchKernelGGL(k_merge_rows, dim3((nsLocal + 255) / 256), dim3(256), 0, ctx->stream, (const RowBest*)rowPart, rsplits, nsLocal, maxErr, pairs, fit);
		BHIP_HIP(ctx, hipGetLastError());
		if (colTop) {
			BHIP_TRY(bhip_ham_mfma_scan(ctx, true, dstB, dstP, nd, srcB, srcP, nsLocal, 16, srcBegin, maxErr, colPart, csplits));
			hipLaunchKernelGGL(k_merge_cols, dim3((nd + 255) / 256), dim3(256), 0, ctx->stream, (const ColTop*)colPart, csplits, nd, colTop);
			BHIP_HIP(ctx, hipGetLastError());
		}
		return BHIP_OK;
	}
	if (hamming) BHIP_TRY((scanHam<false>(ctx, (const int*)src, nsLocal, (const int*)dst, nd, len, 0, maxErr, rowPart, rsplits)));
	else BHIP_TRY((scanL2<false>(ctx, (const double*)src, nsLocal, (const double*)dst, nd, len, 0, maxErr, sqrtScore, rowPart, rsplits)));
	hipLaunchKernelGGL(k_merge_rows, dim3((nsLocal + 255) / 256), dim3(256), 0, ctx->stream, (const RowBest*)rowPart, rsplits, nsLocal, maxErr, pairs, fit);
	BHIP_HIP(ctx, hipGetLastError());
	if (colTop) {
		if (hamming) BHIP_TRY((scanHam<true>(ctx, (const int*)dst, nd, (const int*)src, nsLocal, len, srcBegin, maxErr, colPart, csplits)));
		else BHIP_TRY((scanL2<true>(ctx, (const double*)dst, nd, (const double*)src, nsLocal, len, srcBegin, maxErr, sqrtScore, colPart, csplits)));
		hipLaunchKernelGGL(k_merge_cols, dim3((nd + 255) / 256), dim3(256), 0, ctx->stream, (const ColTop*)colPart, csplits, nd, colTop);
		BHIP_HIP(ctx, hipGetLastError());
	}
	return BHIP_OK;
}

static int phase2(bhip_ctx* ctx, const ColTop* colAll, int nranks, int nd, int nsLocal, int srcBegin, int* pairs, double* fit) {
	if (nsLocal <= 0 || nd <= 0) return BHIP_OK;
	hipLaunchKernelGGL(k_assoc_finish, dim3((nsLocal + 255) / 256), dim3(256), 0, ctx->stream, colAll, nranks, nd, nsLocal, srcBegin, pairs, fit);
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Batched Hamming association: many small independent problems (frame i against frame i+1 of a batch of BRIEF word lists) in three
// launches.  A problem of ~2000 x 2000 x 512 bits is far too small to fill the chip on its own (and the int8 matrix-core path pays a byte
// expansion per call), so here one thread owns a word list of U in registers and scans every word list of V through an LDS tile with
// xor + popcount: exact integer scores, the reference's rules applied directly --
//   rows   : minimum score, LARGEST destination index among equal minima (`fit <= best`), inclusive maxFitError
//   columns: (min1, argmin1, min2) over the sources; (i -> m) survives iff i is the arg-min of column m and min2 > min1.
// Bound: VALU popcount, 2 ops per word and pair: Ns * Nd * words * 2 per pass.
struct HamProb {
	long long srcOff, dstOff, colOff;   // rows into the source / destination word arrays; first column record of this problem
	int ns, nd;
};
struct HamCol { int min1, min2, idx1; };

template <int WORDS, bool COLMODE>
__global__ __launch_bounds__(256) void k_ham_batched(const int* __restrict__ S, const int* __restrict__ D, const HamProb* __restrict__ probs, int words, int thr,
													  double maxErr, int* __restrict__ pairs, double* __restrict__ fit, HamCol* __restrict__ col) {
	__shared__ int tile[VT * (WORDS > 0 ? WORDS : 64)];
	const HamProb pr = probs[blockIdx.y];
	const int W = WORDS > 0 ? WORDS : words;
	const int nU = COLMODE ? pr.nd : pr.ns, nV = COLMODE ? pr.ns : pr.nd;
	if ((int)(blockIdx.x * blockDim.x) >= nU) return;   // whole block: uniform
	const int* __restrict__ U = COLMODE ? D + pr.dstOff * W : S + pr.srcOff * W;
	const int* __restrict__ V = COLMODE ? S + pr.srcOff * W : D + pr.dstOff * W;
	const int ui = blockIdx.x * blockDim.x + threadIdx.x;
	const bool active = ui < nU;
	int a[WORDS > 0 ? WORDS : 1];
	const int* urow = U + (long long)(active ? ui : 0) * W;
	if (WORDS > 0) {
#pragma unroll
		for (int k = 0; k < WORDS; k++) a[k] = urow[k];
	}
	int best = thr, bestIdx = -1;                 // rows: running bound starts at floor(maxFitError)
	int min1 = 0x7fffffff, min2 = 0x7fffffff;     // columns
	for (int t0 = 0; t0 < nV; t0 += VT) {
		const int nt = min(VT, nV - t0);
		__syncthreads();
		for (int e = threadIdx.x; e < nt * W; e += blockDim.x) tile[e] = V[(long long)t0 * W + e];
		__syncthreads();
		if (active) {
			for (int j = 0; j < nt; j++) {
				int sc = 0;
				if (WORDS > 0) {
#pragma unroll
					for (int k = 0; k < WORDS; k++) sc += __popc((unsigned)(a[k] ^ tile[j * WORDS + k]));
				} else {
					for (int k = 0; k < W; k++) sc += __popc((unsigned)(urow[k] ^ tile[j * W + k]));
				}
				if (!COLMODE) {
					if (sc <= best) { best = sc; bestIdx = t0 + j; }
				} else {
					if (sc < min1) { min2 = min1; min1 = sc; bestIdx = t0 + j; }
					else if (sc < min2) { min2 = sc; }
				}
			}
		}
	}
	if (!active) return;
	if (!COLMODE) {
		pairs[pr.srcOff + ui] = bestIdx;
		fit[pr.srcOff + ui] = bestIdx >= 0 ? (double)best : maxErr;
	} else {
		HamCol c; c.min1 = min1; c.min2 = min2; c.idx1 = bestIdx;
		col[pr.colOff + ui] = c;
	}
}
__global__ __launch_bounds__(256) void k_ham_batched_finish(const HamProb* __restrict__ probs, const HamCol* __restrict__ col, int* __restrict__ pairs,
															 double* __restrict__ fit) {
	const HamProb pr = probs[blockIdx.y];
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= pr.ns) return;
	const int m = pairs[pr.srcOff + i];
	if (m < 0) return;
	const HamCol c = col[pr.colOff + m];
	if (!(c.idx1 == i && c.min2 > c.min1)) {
		pairs[pr.srcOff + i] = -1;
		fit[pr.srcOff + i] = DBL_MAX;
	}
}

// problems: host table (count entries; colOff filled in here).  work: scratch for the table and the column records.
int bhip_assoc_hamming_batched(bhip_ctx* ctx, const int32_t* src, const int32_t* dst, int words, int count, const long long* srcOff, const int* ns,
							   const long long* dstOff, const int* nd, double maxErr, int backwards, int* pairs, double* fit, DevBuf& work) {
	if (count <= 0) return BHIP_OK;
	if (words <= 0 || words > 64) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "BRIEF descriptor too long for the batched Hamming kernel");
	if ((size_t)count * sizeof(HamProb) > (1u << 20)) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "too many problems in one batched call");
	HamProb* hp = (HamProb*)ctx->hostScratch;   // pinned: the copy below is asynchronous
	long long cols = 0;
	int maxNs = 0, maxNd = 0;
	for (int p = 0; p < count; p++) {
		if (ns[p] < 0 || nd[p] < 0) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad sizes");
		hp[p].srcOff = srcOff[p]; hp[p].dstOff = dstOff[p]; hp[p].colOff = cols; hp[p].ns = ns[p]; hp[p].nd = nd[p];
		cols += nd[p];
		maxNs = std::max(maxNs, ns[p]); maxNd = std::max(maxNd, nd[p]);
	}
	if (maxNs == 0) return BHIP_OK;
	const size_t tabBytes = ((size_t)count * sizeof(HamProb) + 255) & ~(size_t)255;
	BHIP_TRY(work.reserve(ctx, tabBytes + (size_t)std::max<long long>(cols, 1) * sizeof(HamCol)));
	HamProb* dp = (HamProb*)work.p;
	HamCol* dcol = (HamCol*)((char*)work.p + tabBytes);
	BHIP_HIP(ctx, hipMemcpyAsync(dp, hp, (size_t)count * sizeof(HamProb), hipMemcpyHostToDevice, ctx->stream));
	// integer scores: `score <= maxFitError` is `score <= floor(maxFitError)`; a negative or NaN bound admits nothing
	int thr = -1;
	if (maxErr >= 0) thr = maxErr >= 2147483647.0 ? 0x7fffffff : (int)floor(maxErr);
	const double ops = 2.0 * words;
	double pairsTotal = 0;
	for (int p = 0; p < count; p++) pairsTotal += (double)ns[p] * nd[p];
	{
		ProfScope ps(ctx, "k_ham_batched_rows", 0, ops * pairsTotal);
		const dim3 grid((maxNs + 255) / 256, count);
		if (words == 16) hipLaunchKernelGGL((k_ham_batched<16, false>), grid, dim3(256), 0, ctx->stream, src, dst, dp, words, thr, maxErr, pairs, fit, dcol);
		else hipLaunchKernelGGL((k_ham_batched<0, false>), grid, dim3(256), 0, ctx->stream, src, dst, dp, words, thr, maxErr, pairs, fit, dcol);
		BHIP_HIP(ctx, hipGetLastError());
	}
	if (backwards && maxNd > 0) {
		{
			ProfScope ps(ctx, "k_ham_batched_cols", 0, ops * pairsTotal);
			const dim3 grid((maxNd + 255) / 256, count);
			if (words == 16) hipLaunchKernelGGL((k_ham_batched<16, true>), grid, dim3(256), 0, ctx->stream, src, dst, dp, words, thr, maxErr, pairs, fit, dcol);
			else hipLaunchKernelGGL((k_ham_batched<0, true>), grid, dim3(256), 0, ctx->stream, src, dst, dp, words, thr, maxErr, pairs, fit, dcol);
			BHIP_HIP(ctx, hipGetLastError());
		}
		hipLaunchKernelGGL(k_ham_batched_finish, dim3((maxNs + 255) / 256, count), dim3(256), 0, ctx->stream, dp, dcol, pairs, fit);
		BHIP_HIP(ctx, hipGetLastError());
	}
	// the pinned table is reused by the next call on this ctx: the copy must have been consumed
	BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return BHIP_OK;
}

// ---- entry points used by cabi.cpp ----
int bhip_assoc_phase1_l2(bhip_ctx* ctx, const double* src, int nsLocal, int srcBegin, const double* dst, int nd, int dof, double maxErr, int sqrtScore,
						 int* pairs, double* fit, void* colTop, DevBuf& work) {
	return phase1<double>(ctx, false, src, nsLocal, srcBegin, dst, nd, dof, maxErr, sqrtScore, pairs, fit, (ColTop*)colTop, work);
}
int bhip_assoc_phase1_ham(bhip_ctx* ctx, const int32_t* src, int nsLocal, int srcBegin, const int32_t* dst, int nd, int words, double maxErr, int* pairs,
						  double* fit, void* colTop, DevBuf& work) {
	return phase1<int>(ctx, true, (const int*)src, nsLocal, srcBegin, (const int*)dst, nd, words, maxErr, 0, pairs, fit, (ColTop*)colTop, work);
}
int bhip_assoc_phase2(bhip_ctx* ctx, const void* colAll, int nranks, int nd, int nsLocal, int srcBegin, int* pairs, double* fit) {
	return phase2(ctx, (const ColTop*)colAll, nranks, nd, nsLocal, srcBegin, pairs, fit);
}
int bhip_assoc_coltop_size() { return (int)sizeof(ColTop); }
