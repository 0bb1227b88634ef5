// K1: integral image (summed-area table), bit-exact with the reference's sequential fp32 order.
//
// Reference: ImplIntegralImageOps.transform(GrayF32,GrayF32)  I:alg/transform/ii/impl/ImplIntegralImageOps.java:42-66
//   row y:  t = 0; for x: t += p[y][x];  ii[y][x] = ii[y-1][x] + t        (ii[-1][x] absent for y = 0)
// fp32 addition is not associative, so both chains stay strictly sequential:
//   pass 1 (rows):    s[y][x] = (((p[y][0] + p[y][1]) + ...) + p[y][x])     one chain per row, parallel over rows x batch
//   pass 2 (columns): ii[y][x] = ((s[0][x] + s[1][x]) + ...) + s[y][x]      one chain per column, parallel over columns x batch
// Note ii[0][x] = s[0][x] exactly (the reference stores the running total for row 0).
//
// Memory: pass 1 reads rows through an LDS transpose so that HBM accesses stay coalesced (a wave reads 256 B row segments)
// while every lane owns one row's chain; pass 2 is naturally coalesced (a wave owns 64 adjacent columns).
// Bound: HBM.  Algorithmic bytes: 4P read + 4P write per pass.
#include "common.h"
#include <cstdlib>

#define TILE 64
#define TPAD 65   // (row*65 + k) % 32 == (row + k) % 32: conflict-free column walks for ds_read/write_b32

// one wave = one 64-row group; it sweeps the rows left to right in 64-column tiles.  Row -> (image, y) is tracked with wave-uniform
// counters (no per-load division); the next tile's 64 row segments are loaded into registers while the current tile is scanned.
__global__ __launch_bounds__(256) void k_integral_rows(ImgView in, ImgViewW out, long long totalRows) {
	__shared__ float lds[4][TILE * TPAD];
	const int wave = threadIdx.x >> 6;
	const int lane = threadIdx.x & 63;
	float* tile = lds[wave];
	const long long row0 = ((long long)blockIdx.x * 4 + wave) * TILE;
	if (row0 >= totalRows) return;
	const int H = in.height, W = in.width;
	const int nrows = (int)min((long long)TILE, totalRows - row0);
	const long long img0 = row0 / H;
	const int y0 = (int)(row0 - img0 * H);
	const bool myRowValid = lane < nrows;
	float carry = 0.0f;

	float v[TILE];
	// prologue: first tile
	{
		long long img = img0; int y = y0;
#pragma unroll
		for (int r = 0; r < TILE; r++) {
			v[r] = (r < nrows && lane < W) ? in.data[img * in.imageStride + (long long)y * in.stride + lane] : 0.0f;
			if (++y == H) { y = 0; img++; }
		}
	}
	for (int c0 = 0; c0 < W; c0 += TILE) {
		const int x = c0 + lane;
#pragma unroll
		for (int r = 0; r < TILE; r++) tile[r * TPAD + lane] = v[r];
		// issue the next tile's loads now; they complete under the scan below
		if (c0 + TILE < W) {
			const int xn = x + TILE;
			long long img = img0; int y = y0;
#pragma unroll
			for (int r = 0; r < TILE; r++) {
				v[r] = (r < nrows && xn < W) ? in.data[img * in.imageStride + (long long)y * in.stride + xn] : 0.0f;
				if (++y == H) { y = 0; img++; }
			}
		}
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		// sequential prefix along the row: lane = row
		if (myRowValid) {
			float* mine = tile + lane * TPAD;
			const int n = min(TILE, W - c0);
			if (n == TILE) {
#pragma unroll
				for (int k0 = 0; k0 < TILE; k0 += 16) {
					float t[16];
#pragma unroll
					for (int k = 0; k < 16; k++) t[k] = mine[k0 + k];
#pragma unroll
					for (int k = 0; k < 16; k++) { carry += t[k]; t[k] = carry; }
#pragma unroll
					for (int k = 0; k < 16; k++) mine[k0 + k] = t[k];
				}
			} else {
				for (int k = 0; k < n; k++) {
					carry += mine[k];
					mine[k] = carry;
				}
			}
		}
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		{
			long long img = img0; int y = y0;
#pragma unroll
			for (int r = 0; r < TILE; r++) {
				if (r < nrows && x < W) out.data[img * out.imageStride + (long long)y * out.stride + x] = tile[r * TPAD + lane];
				if (++y == H) { y = 0; img++; }
			}
		}
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
	}
}

// in-place column accumulation: thread = one column of one image
__global__ __launch_bounds__(256) void k_integral_cols(ImgViewW io) {
	const int x = blockIdx.x * blockDim.x + threadIdx.x;
	if (x >= io.width) return;
	float* p = io.data + (long long)blockIdx.y * io.imageStride + x;
	const int H = io.height;
	const long long s = io.stride;
	float acc = p[0];
	int y = 1;
	for (; y + 8 <= H; y += 8) {
		float v[8];
#pragma unroll
		for (int k = 0; k < 8; k++) v[k] = p[(y + k) * s];
#pragma unroll
		for (int k = 0; k < 8; k++) {
			acc = acc + v[k];
			p[(y + k) * s] = acc;
		}
	}
	for (; y < H; y++) {
		acc = acc + p[y * s];
		p[y * s] = acc;
	}
}

// ---------------------------------------------------------------------------------------------------------------
// Single-pass form for large batches: one workgroup (16 waves) per image, the input read once and the integral image written once
// (8P bytes instead of 16P).  Both chains stay sequential:
//   * a wave owns a band of 64 rows and walks it left to right in 64x32 tiles; lane = row keeps that row's running sum in a register
//     across the tiles (the row chain), exactly as k_integral_rows does;
//   * inside a tile lane = column then adds the band's row sums on top of the integral-image row just above the band (the column
//     chain).  That row travels through LDS: the wave owning the band above left it in a two-slot ring one step earlier; across the
//     wrap from wave 15 to wave 0 (images with more than 16 bands) it waits in a full-width row buffer.
// Band b starts one step after band b - 1 (a diagonal pipeline over the 16 waves); with more than 16 bands a wave takes band b + 16
// when it has finished band b.  Every wave runs the same number of steps and meets one barrier per step, whether it has a tile or not.
// Step of (band b, tile j): (b / 16) * max(nTiles, 16) + b % 16 + j.  The barrier orders LDS only (s_waitcnt lgkmcnt(0) + s_barrier):
// nothing read by another wave goes through global memory, so stores and the next tile's loads stay in flight across steps.
// ---------------------------------------------------------------------------------------------------------------
#define FT 32   // tile columns
#define FP 33   // LDS pitch: (row * 33 + k) % 32 == (row + k) % 32
#define FUSED_RING_FLOATS (2 * 16 * FT)
#define FUSED_MIN_BATCH 128   // an image takes ~0.8 ms through its one workgroup whatever the batch (pipeline steps of ~7 us, set by
                              // memory latency with one tile of look-ahead); the two streaming passes are faster below ~128 images
                              // (1080p: 0.72 vs 0.84 ms at 96, 0.96 vs 0.85 at 128, 1.81 vs 1.08-1.16 at 256)

// BH = rows per band: 64, 68 or 72 (even; the tiles of the 16 waves have to fit the LDS).  A band taller than 64 rows gives lanes
// 0 .. BH - 65 a second row (rows 64 .. BH - 1) with a row chain of its own.  The launch picks the height that needs the fewest rounds of 16
// bands: a 1080-row frame is 16 bands of 68 rows -- one round, every wave busy -- where 64-row bands leave a 17th band to run alone for a
// whole second round.
template <int BH>
__global__ __launch_bounds__(1024) void k_integral_fused(ImgView in, ImgViewW out) {
	static_assert(BH >= 64 && BH <= 128 && (BH & 1) == 0, "band height");
	extern __shared__ float fusedLds[];   // [16][BH * FP] tiles | [2][16][FT] ring | [W] wrap row (only with more than 16 bands)
	const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
	float* tile = fusedLds + wave * BH * FP;
	float* ring = fusedLds + 16 * BH * FP;
	float* wrapRow = ring + FUSED_RING_FLOATS;
	const int half = lane >> 5, c = lane & 31;
	const int W = in.width, H = in.height;
	const float* __restrict__ src = in.data + (long long)blockIdx.x * in.imageStride;
	float* __restrict__ dst = out.data + (long long)blockIdx.x * out.imageStride;
	const int nTiles = (W + FT - 1) / FT;
	const int nBands = (H + BH - 1) / BH;
	const int rounds = (nBands + 15) / 16;
	const int S = max(nTiles, 16);
	const int T = (rounds - 1) * S + 16 + nTiles;   // steps; identical for every wave

	// (band, tile) of this wave at step t, or band = -1
	auto slot = [&](int t, int& b, int& j) {
		const int u = t - wave;
		b = -1; j = 0;
		if (u < 0) return;
		const int r = u / S;
		j = u - r * S;
		const int bb = wave + 16 * r;
		if (r < rounds && bb < nBands && j < nTiles) b = bb;
	};
	// Full tiles (BH rows, 32 columns inside the image: all but the last band / last tile) take a path without per-element guards and
	// with scalar row bases + one 32-bit lane offset, so a load or store costs one instruction; the guarded form handles the edges.
	const unsigned laneOffIn = (unsigned)(half * in.stride + c), laneOffOut = (unsigned)(half * out.stride + c);
	auto loadTile = [&](int b, int j, float* v) {
		const int y0 = BH * b;
		const int nrows = min(BH, H - y0);
		if (FT * j + FT <= W) {
			const float* __restrict__ base = src + (long long)y0 * in.stride + FT * j;   // wave-uniform
			if (nrows == BH) {
#pragma unroll
				for (int i = 0; i < BH / 2; i++) v[i] = (base + (long long)(2 * i) * in.stride)[laneOffIn];
			} else {
				// last band: the row tests are wave-uniform (scalar branches), only an odd last row needs a lane predicate
#pragma unroll
				for (int i = 0; i < BH / 2; i++) {
					const float* rp = base + (long long)(2 * i) * in.stride;
					if (2 * i + 1 < nrows) v[i] = rp[laneOffIn];
					else if (2 * i < nrows) v[i] = half == 0 ? rp[laneOffIn] : 0.0f;
					else v[i] = 0.0f;
				}
			}
			return;
		}
		const int col = FT * j + c;
#pragma unroll
		for (int i = 0; i < BH / 2; i++) {
			const int row = 2 * i + half;
			v[i] = (row < nrows && col < W) ? src[(long long)(y0 + row) * in.stride + col] : 0.0f;
		}
	};
	float v[BH / 2];
	float carry = 0.0f, carry2 = 0.0f;   // row chains of rows `lane` and `lane + 64`
	{
		int b, j;
		slot(0, b, j);
		if (b >= 0) loadTile(b, j, v);
	}
	for (int t = 0; t < T; t++) {
		int b, j;
		slot(t, b, j);
		if (b >= 0) {
			const int y0 = BH * b;
			const int nrows = min(BH, H - y0);
			if (j == 0) { carry = 0.0f; carry2 = 0.0f; }
			const int colC = FT * j + c;
			// the integral-image row above this band, for the tile's columns
			float top = 0.0f;
			if (b > 0 && half == 0) top = wave > 0 ? ring[((t - 1) & 1) * 16 * FT + (wave - 1) * FT + c] : wrapRow[min(colC, W - 1)];
#pragma unroll
			for (int i = 0; i < BH / 2; i++) tile[(2 * i + half) * FP + c] = v[i];
			// next step's tile: its loads complete under this step's scans
			{
				int nb, nj;
				slot(t + 1, nb, nj);
				if (nb >= 0) loadTile(nb, nj, v);
			}
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			// row chain: lane = row (and row lane + 64 of a taller band)
			if (lane < nrows) {
				float* mine = tile + lane * FP;
#pragma unroll
				for (int k0 = 0; k0 < FT; k0 += 16) {
					float q[16];
#pragma unroll
					for (int k = 0; k < 16; k++) q[k] = mine[k0 + k];
#pragma unroll
					for (int k = 0; k < 16; k++) { carry += q[k]; q[k] = carry; }   // columns past W hold 0: harmless, never stored
#pragma unroll
					for (int k = 0; k < 16; k++) mine[k0 + k] = q[k];
				}
			}
			if (BH > 64 && lane + 64 < nrows) {
				float* mine = tile + (lane + 64) * FP;
#pragma unroll
				for (int k0 = 0; k0 < FT; k0 += 16) {
					float q[16];
#pragma unroll
					for (int k = 0; k < 16; k++) q[k] = mine[k0 + k];
#pragma unroll
					for (int k = 0; k < 16; k++) { carry2 += q[k]; q[k] = carry2; }
#pragma unroll
					for (int k = 0; k < 16; k++) mine[k0 + k] = q[k];
				}
			}
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			// column chain: lane = column (first half wave); ii[y] = ii[y-1] + s[y], and ii[0] = s[0] on the image's first row
			if (half == 0) {
				float acc = top;
				int r0 = 0;
				if (b == 0) { acc = tile[c]; r0 = 1; }
				// 16 rows at a time: the LDS reads of a batch are issued together, then the dependent adds, then the writes
#pragma unroll
				for (int rb = 0; rb < BH; rb += 16) {
					if (rb < nrows) {
						float q[16];
#pragma unroll
						for (int k = 0; k < 16; k++) q[k] = rb + k < BH ? tile[(rb + k) * FP + c] : 0.0f;   // rows past nrows: guarded below
#pragma unroll
						for (int k = 0; k < 16; k++)
							if (rb + k < BH && rb + k >= r0 && rb + k < nrows) { acc = acc + q[k]; q[k] = acc; }
#pragma unroll
						for (int k = 0; k < 16; k++)
							if (rb + k < BH && rb + k >= r0 && rb + k < nrows) tile[(rb + k) * FP + c] = q[k];
					}
				}
				// the band's last row for the band below: next wave's ring slot, or the wrap row when that band belongs to wave 0
				if (b + 1 < nBands) {
					if (wave < 15) ring[(t & 1) * 16 * FT + wave * FT + c] = acc;
					else if (colC < W) wrapRow[colC] = acc;
				}
			}
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
			if (FT * j + FT <= W) {
				float* __restrict__ base = dst + (long long)y0 * out.stride + FT * j;   // wave-uniform
				if (nrows == BH) {
#pragma unroll
					for (int i = 0; i < BH / 2; i++) (base + (long long)(2 * i) * out.stride)[laneOffOut] = tile[(2 * i + half) * FP + c];
				} else {
#pragma unroll
					for (int i = 0; i < BH / 2; i++) {
						float* rp = base + (long long)(2 * i) * out.stride;
						if (2 * i + 1 < nrows) rp[laneOffOut] = tile[(2 * i + half) * FP + c];
						else if (2 * i < nrows && half == 0) rp[laneOffOut] = tile[(2 * i) * FP + c];
					}
				}
			} else {
#pragma unroll
				for (int i = 0; i < BH / 2; i++) {
					const int row = 2 * i + half;
					if (row < nrows && colC < W) dst[(long long)(y0 + row) * out.stride + colC] = tile[row * FP + c];
				}
			}
		} else {
			int nb, nj;
			slot(t + 1, nb, nj);
			if (nb >= 0) loadTile(nb, nj, v);
		}
		// LDS-only barrier: ring / wrap-row writes of this step are complete and visible; global loads and stores are not waited for
		asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
	}
}

// band height of the single-pass kernel for an H-row image: fewest rounds of 16 bands, then the shorter band
static int fusedBandHeight(int H) {
	int best = 64, bestRounds = 1 << 30;
	for (int bh = 64; bh <= 72; bh += 4) {
		const int rounds = ((H + bh - 1) / bh + 15) / 16;
		if (rounds < bestRounds) { bestRounds = rounds; best = bh; }
	}
	return best;
}

int bhip_launch_integral(bhip_ctx* ctx, ImgView in, ImgViewW out, int batch) {
	if (in.width <= 0 || in.height <= 0 || batch <= 0) return BHIP_OK;
	const bool twoPass = bhip_env_flag("BHIP_INTEGRAL_TWO_PASS");   // parity cross-check of the two integral plans
	// one workgroup per image only fills the chip with a large batch; small batches keep the two streaming passes
	int bh = fusedBandHeight(in.height);
	{ const char* e = getenv("BHIP_INTEGRAL_BAND"); if (e && (atoi(e) == 64 || atoi(e) == 68 || atoi(e) == 72)) bh = atoi(e); }   // parity cross-check of the band heights
	const bool wrap = (in.height + bh - 1) / bh > 16;
	const size_t lds = ((size_t)16 * bh * FP + FUSED_RING_FLOATS + (wrap ? in.width : 0)) * sizeof(float);
	if (!twoPass && batch >= FUSED_MIN_BATCH && in.data != out.data && lds <= 160 * 1024) {
		// the attribute is per device and this library serves one ctx per (thread, device): remember it in the ctx, not in a static
		if (lds > ctx->integralLdsAttr) {
			BHIP_HIP(ctx, hipFuncSetAttribute((const void*)k_integral_fused<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
			BHIP_HIP(ctx, hipFuncSetAttribute((const void*)k_integral_fused<68>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
			BHIP_HIP(ctx, hipFuncSetAttribute((const void*)k_integral_fused<72>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
			ctx->integralLdsAttr = lds;
		}
		ProfScope ps(ctx, "k_integral_fused", 8.0 * in.width * in.height * batch);   // 4P read + 4P write
		if (bh == 64) hipLaunchKernelGGL(k_integral_fused<64>, dim3(batch), dim3(1024), lds, ctx->stream, in, out);
		else if (bh == 68) hipLaunchKernelGGL(k_integral_fused<68>, dim3(batch), dim3(1024), lds, ctx->stream, in, out);
		else hipLaunchKernelGGL(k_integral_fused<72>, dim3(batch), dim3(1024), lds, ctx->stream, in, out);
		BHIP_HIP(ctx, hipGetLastError());
		return BHIP_OK;
	}
	const long long totalRows = (long long)in.height * batch;
	const long long groups = (totalRows + TILE - 1) / TILE;
	const unsigned blocks = (unsigned)((groups + 3) / 4);
	const double passBytes = 8.0 * in.width * in.height * batch;  // 4P read + 4P write per pass
	{
		ProfScope ps(ctx, "k_integral_rows", passBytes);
		hipLaunchKernelGGL(k_integral_rows, dim3(blocks), dim3(256), 0, ctx->stream, in, out, totalRows);
	}
	dim3 grid((in.width + 255) / 256, batch);
	{
		ProfScope ps(ctx, "k_integral_cols", passBytes);
		hipLaunchKernelGGL(k_integral_cols, grid, dim3(256), 0, ctx->stream, out);
	}
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}
