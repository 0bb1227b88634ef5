// K7 (MFMA path): greedy Hamming association on the int8 matrix cores -- exact.
//
// Reference: AssociateGreedy.associate   F:alg/feature/associate/AssociateGreedy.java:65-118 with
//            DescriptorDistance.hamming  F:alg/descriptor/DescriptorDistance.java:196-220  ("{0,1} int8 GEMM, exact in int32")
//
// A descriptor of W 32-bit words is expanded once per call to 32W bytes of {0,1}.  Then
//     hamming(a, b) = popcount(a) + popcount(b) - 2 <a, b>
// and the inner products of a 32 x 32 tile are W v_mfma_i32_32x32x32_i8 instructions with int32 accumulation: exact, so the
// reference's rules are applied directly to the integer scores (no candidate lists, no re-scoring):
//   rows   : minimum score, LARGEST destination index among equal minima (`fit <= best` of the forward loop), inclusive maxFitError
//   columns: (min1, argmin1, min2) over the sources, for the strict mutual-best test / the sharded all-gather records
// Outputs are the same per-split partial records the VALU scan kernel writes (RowBest / ColTop), so merging, the sharded phase 2 and
// the tie handling stay in associate.hip.
//
// Operand map: lane l (r = l & 31, h = l >> 5) holds bytes [32 s + 16 h, +16) of row r for k-step s as BOTH operands' fragment.  Any
// split of the 32 k of a step between the two lane halves gives the same sums as long as A and B use the same one (checked with exact
// integer data, scripts/probe/mfma_i8_layout.hip).  C layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
// Bound: MFMA (int8 dense rate); algorithmic ops per pass = 2 * Ns * Nd * 32W.
#include "common.h"
#include <cfloat>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

struct HamColTop {   // must match ColTop of associate.hip
	double min1, min2;
	int idx1, pad;
};
struct HamRowBest {  // must match RowBest of associate.hip
	double best;
	int idx, pad;
};

// one thread per (row, word): 32 bits -> 32 bytes of {0,1}; one thread per row for the popcount
__global__ __launch_bounds__(256) void k_ham_expand(const int* __restrict__ D, long long rows, int words, unsigned char* __restrict__ bytes, int* __restrict__ pop) {
	const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= rows * words) return;
	const unsigned int x = (unsigned int)D[t];
	unsigned int out[8];
#pragma unroll
	for (int q = 0; q < 8; q++) {
		const unsigned int n = (x >> (4 * q)) & 15u;
		out[q] = (n & 1u) | ((n & 2u) << 7) | ((n & 4u) << 14) | ((n & 8u) << 21);
	}
	uint4* dst = (uint4*)(bytes + t * 32);
	dst[0] = make_uint4(out[0], out[1], out[2], out[3]);
	dst[1] = make_uint4(out[4], out[5], out[6], out[7]);
	if (t % words == 0) {
		int s = 0;
		for (int k = 0; k < words; k++) s += __popc((unsigned int)D[t + k]);
		pop[t / words] = s;
	}
}

// Each wave owns 32 rows of U (fragments in registers); the four waves of a block sweep the same split of V, 32 columns at a time, from
// ONE copy of the tile in LDS (the byte expansion makes the operands 8x larger than the bit strings: without sharing the sweep is
// bound by L2 traffic, 16 KB per 16 MFMAs).  The tile is stored as [k-step][lane half][column] 16-byte chunks with a pitch of 33 chunks:
// fragment reads are contiguous across lanes, staging writes spread over the banks.  Next tile's chunks are fetched into registers
// before the MFMAs of the current one (double buffer).
//   COLMODE = false: U = sources, V = destinations -> HamRowBest partial out[split][u]
//   COLMODE = true : U = destinations, V = sources -> HamColTop partial out[split][u], idx = vBase + v
template <int WORDS, bool COLMODE>
__global__ __launch_bounds__(256, 2) void k_ham_mfma(const unsigned char* __restrict__ Ub, const int* __restrict__ Up, int nU, const unsigned char* __restrict__ Vb,
													  const int* __restrict__ Vp, int nV, int vPerSplit, int vBase, double maxErr, void* __restrict__ outRaw) {
	constexpr int KB = 32 * WORDS;          // bytes per expanded row
	constexpr int CH = 2 * WORDS;           // 16-byte chunks per row
	constexpr int NCH = 32 * CH;            // chunks per 32-column tile
	constexpr int PER_T = (NCH + 255) / 256;
	__shared__ uint4 tile[2][CH * 33];
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	const int r = lane & 31, h = lane >> 5;
	const int uT = (blockIdx.x * 4 + wave) * 32;
	const bool live = uT < nU;              // a dead wave still takes part in staging and barriers
	const int split = blockIdx.y;
	const int v0 = split * vPerSplit, v1 = min(nV, v0 + vPerSplit);

	v4i a[WORDS];
	{
		const int row = min(uT + r, nU - 1);   // rows past the end repeat the last one and are not written
		const v4i* src = (const v4i*)(Ub + (long long)row * KB + 16 * h);
#pragma unroll
		for (int s = 0; s < WORDS; s++) a[s] = src[2 * s];
	}
	int pu[16];
#pragma unroll
	for (int g = 0; g < 16; g++) pu[g] = Up[min(uT + (g & 3) + 8 * (g >> 2) + 4 * h, nU - 1)];

	unsigned int k1[16];   // rows: (score << 16) | (0xFFFF - local column)   columns: (score << 16) | local source
	int m2[16];            // columns only: second smallest score
#pragma unroll
	for (int g = 0; g < 16; g++) { k1[g] = 0xFFFFFFFFu; m2[g] = 0xFFFF; }

	// chunk q of this thread in a tile: global chunk index g = tid + 256 q -> column g / CH, chunk-in-row g % CH
	uint4 pre[PER_T];
	auto fetch = [&](int c0) {
#pragma unroll
		for (int q = 0; q < PER_T; q++) {
			const int g = tid + 256 * q;
			const int col = c0 + g / CH;
			pre[q] = (g < NCH && col < v1) ? ((const uint4*)(Vb + (long long)col * KB))[g % CH] : make_uint4(0, 0, 0, 0);
		}
	};
	auto stash = [&](int buf) {
#pragma unroll
		for (int q = 0; q < PER_T; q++) {
			const int g = tid + 256 * q;
			if (g < NCH) tile[buf][(g % CH) * 33 + g / CH] = pre[q];
		}
	};
	fetch(v0);
	stash(0);
	__syncthreads();
	int buf = 0;
	for (int c0 = v0; c0 < v1; c0 += 32, buf ^= 1) {
		const bool more = c0 + 32 < v1;
		if (more) fetch(c0 + 32);
		const int col = c0 + r;
		const bool ok = col < v1;
		v4i b[WORDS];
#pragma unroll
		for (int s = 0; s < WORDS; s++) {
			const uint4 t = tile[buf][(2 * s + h) * 33 + r];
			b[s] = v4i{(int)t.x, (int)t.y, (int)t.z, (int)t.w};
		}
		const int pv = ok ? Vp[col] : 0;
		// two accumulators over alternating k-steps: independent MFMA chains, added exactly afterwards
		v16i accA = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, accB = accA;
#pragma unroll
		for (int s = 0; s < WORDS; s += 2) {
			accA = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s], b[s], accA, 0, 0, 0);
			if (s + 1 < WORDS) accB = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[s + 1], b[s + 1], accB, 0, 0, 0);
		}
		const unsigned int local = (unsigned int)(col - v0);
#pragma unroll
		for (int g = 0; g < 16; g++) {
			const int ham = (pu[g] + pv) - 2 * (accA[g] + accB[g]);
			if (!COLMODE) {
				const unsigned int key = ok ? (((unsigned int)ham << 16) | (0xFFFFu - local)) : 0xFFFFFFFFu;
				k1[g] = min(k1[g], key);
			} else {
				const unsigned int key = ok ? (((unsigned int)ham << 16) | local) : 0xFFFFFFFFu;
				const int hamv = ok ? ham : 0xFFFF;
				if (key < k1[g]) { m2[g] = min(m2[g], (int)(k1[g] >> 16)); k1[g] = key; }
				else m2[g] = min(m2[g], hamv);
			}
		}
		if (more) stash(buf ^ 1);   // the other buffer was last read before the previous barrier
		__syncthreads();
	}
	if (!live) return;
	// combine the 32 lanes of a half (they hold different columns of V for the same 16 rows)
#pragma unroll
	for (int g = 0; g < 16; g++) {
#pragma unroll
		for (int o = 16; o >= 1; o >>= 1) {
			const unsigned int ok1 = (unsigned int)__shfl_xor((int)k1[g], o, 64);
			if (COLMODE) {
				const int om2 = __shfl_xor(m2[g], o, 64);
				m2[g] = min(min(m2[g], om2), (int)(max(k1[g], ok1) >> 16));
			}
			k1[g] = min(k1[g], ok1);
		}
	}
	if (r == 0) {
#pragma unroll
		for (int g = 0; g < 16; g++) {
			const int u = uT + (g & 3) + 8 * (g >> 2) + 4 * h;
			if (u >= nU) continue;
			if (!COLMODE) {
				HamRowBest* out = (HamRowBest*)outRaw + (long long)split * nU + u;
				const double fit = (double)(k1[g] >> 16);
				const bool hit = k1[g] != 0xFFFFFFFFu && fit <= maxErr;
				out->best = hit ? fit : maxErr;
				out->idx = hit ? v0 + (int)(0xFFFFu - (k1[g] & 0xFFFFu)) : -1;
				out->pad = 0;
			} else {
				HamColTop* out = (HamColTop*)outRaw + (long long)split * nU + u;
				const bool any = k1[g] != 0xFFFFFFFFu;
				out->min1 = any ? (double)(k1[g] >> 16) : INFINITY;
				out->min2 = m2[g] != 0xFFFF ? (double)m2[g] : INFINITY;
				out->idx1 = any ? vBase + v0 + (int)(k1[g] & 0xFFFFu) : -1;
				out->pad = 0;
			}
		}
	}
}

int bhip_ham_expand(bhip_ctx* ctx, const int* D, long long rows, int words, unsigned char* bytes, int* pop) {
	if (rows <= 0) return BHIP_OK;
	ProfScope ps(ctx, "k_ham_expand", (double)rows * words * 36);
	hipLaunchKernelGGL(k_ham_expand, dim3((unsigned)((rows * words + 255) / 256)), dim3(256), 0, ctx->stream, D, rows, words, bytes, pop);
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}

// splits of the swept set so that there are enough waves to fill the chip; every split covers fewer than 65536 vectors (16-bit local index)
int bhip_ham_mfma_splits(int nU, int nV) {
	const int uTiles = (nU + 31) / 32;
	int splits = (2048 + uTiles - 1) / uTiles;
	const int minSplits = (nV + 65535) / 65536;
	const int maxSplits = (nV + 31) / 32;
	if (splits < minSplits) splits = minSplits;
	if (splits > maxSplits) splits = maxSplits;
	if (splits < 1) splits = 1;
	return splits;
}

// words must be 16 (BRIEF-512); partial = [splits][nU] records of the same layout as the VALU scan kernel's
int bhip_ham_mfma_scan(bhip_ctx* ctx, bool colMode, const unsigned char* Ub, const int* Up, int nU, const unsigned char* Vb, const int* Vp, int nV, int words,
						 int vBase, double maxErr, void* partial, int splits) {
	if (words != 16) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "the int8 MFMA Hamming path is built for 16-word descriptors");
	int per = (nV + splits - 1) / splits;
	per = ((per + 31) / 32) * 32;
	if (per > 65536) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "split too long for the 16-bit local index");
	dim3 grid((nU + 127) / 128, splits);
	ProfScope ps(ctx, colMode ? "k_ham_mfma_cols" : "k_ham_mfma_rows", 0, 2.0 * nU * (double)nV * 32 * words);
	if (colMode) hipLaunchKernelGGL((k_ham_mfma<16, true>), grid, dim3(256), 0, ctx->stream, Ub, Up, nU, Vb, Vp, nV, per, vBase, maxErr, partial);
	else hipLaunchKernelGGL((k_ham_mfma<16, false>), grid, dim3(256), 0, ctx->stream, Ub, Up, nU, Vb, Vp, nV, per, vBase, maxErr, partial);
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}
