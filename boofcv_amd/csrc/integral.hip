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

int bhip_launch_integral(bhip_ctx* ctx, ImgView in, ImgViewW out, int batch) {
	if (in.width <= 0 || in.height <= 0 || batch <= 0) return BHIP_OK;
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
