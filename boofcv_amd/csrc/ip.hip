// K8-K10: boofcv-ip front-end ops behind the BOverride* hooks (separable convolution, Gaussian blur, Sobel / three-tap gradient)
// and the BRIEF binary descriptor.  All fp32 with the reference's evaluation order and no FMA contraction.
//
// Reference:
//   ConvolveImageNoBorder.horizontal/vertical     I:alg/filter/convolve/ConvolveImageNoBorder.java:53-77
//     unrolled widths 3,5,7,9,11 (first tap assigns)  I:alg/filter/convolve/noborder/ConvolveImageUnrolled_SB_F32_F32.java:50-150,152-180,347-382
//     standard (total = 0 first)                      I:alg/filter/convolve/noborder/ConvolveImageStandard_SB.java:44-104
//   ConvolveImageNormalized.horizontal/vertical   I:alg/filter/convolve/ConvolveImageNormalized.java:48-93
//     border re-normalisation                         I:alg/filter/convolve/normalized/ConvolveNormalized_JustBorder_SB.java:42-145
//     kernel wider than the image                     I:alg/filter/convolve/normalized/ConvolveNormalizedNaive_SB.java:37-84
//   GradientSobel_UnrolledOuter.process_F32_sub   I:alg/filter/derivative/impl/GradientSobel_UnrolledOuter.java:210-
//   GradientThree_Standard.process                I:alg/filter/derivative/impl/GradientThree_Standard.java:40-62
//   border handling                               I:alg/filter/convolve/border/ConvolveJustBorder_General_SB.java:46-175,
//                                                 I:alg/filter/derivative/DerivativeHelperFunctions.java:140-175
//   ImplDescribeBinaryCompare_F32                 F:alg/feature/describe/impl/ImplDescribeBinaryCompare_F32.java:47-101
// Bound: HBM (8P bytes per separable pass, 12P for a gradient); taps are re-read through L1/L2.
#include "common.h"

#define BHIP_MAX_TAPS 255

struct ConvParams {
	const float* in;
	float* out;
	long long inImageStride, outImageStride;   // floats between the images of a batch (blockIdx.z)
	int inStride, outStride, width, height;
	int kw, koff;
	int unrolled;   // first tap assigns instead of adding to 0
	int mode;       // 0 = no border (frame untouched), 1 = normalised border, 2 = normalised naive (kernel wider than image),
	                // 3 = normalised border pixels only (interior untouched: the mean blur fills it with running sums)
	int borderOnly; // general kernel only: the grid covers just the koff + (kw-koff-1) border columns (rows) of the filtered axis
	float k[BHIP_MAX_TAPS];
};

// Interior taps.  The loads of one output are independent: issue them together (a run-time tap loop waits for each load before the next).
// Unrolled widths (ConvolveImageUnrolled_SB_F32_F32 / ConvolveDownNoBorderUnrolled_F32_F32): total = s[0]*k[0]; total += s[i]*k[i] ...
template <int KW>
__device__ __forceinline__ float tapsFirstAssigns(const float* __restrict__ s, long long step, const float* k) {
	float v[KW];
#pragma unroll
	for (int i = 0; i < KW; i++) v[i] = s[i * step];
	float total = v[0] * k[0];
#pragma unroll
	for (int i = 1; i < KW; i++) total += v[i] * k[i];
	return total;
}
__device__ __forceinline__ float tapsUnrolled(const float* __restrict__ s, long long step, const float* k, int kw) {
	switch (kw) {
	case 3: return tapsFirstAssigns<3>(s, step, k);
	case 5: return tapsFirstAssigns<5>(s, step, k);
	case 7: return tapsFirstAssigns<7>(s, step, k);
	case 9: return tapsFirstAssigns<9>(s, step, k);
	default: return tapsFirstAssigns<11>(s, step, k);
	}
}
// Standard form (ConvolveImageStandard_SB / ConvolveDownNoBorderStandard): total = 0; total += s[i]*k[i] in order, loads four at a time
__device__ __forceinline__ float tapsStandard(const float* __restrict__ s, long long step, const float* k, int kw) {
	float total = 0;
	int i = 0;
	for (; i + 4 <= kw; i += 4) {
		const float v0 = s[i * step], v1 = s[(i + 1) * step], v2 = s[(i + 2) * step], v3 = s[(i + 3) * step];
		total += v0 * k[i];
		total += v1 * k[i + 1];
		total += v2 * k[i + 2];
		total += v3 * k[i + 3];
	}
	for (; i < kw; i++) total += s[i * step] * k[i];
	return total;
}

// General form: one output pixel per thread, taps through L1 / L2.  Any base address, stride and width (sub-images with odd strides,
// kernels wider than the image); the tiled kernels below take over whenever rows are 16-byte aligned.
template <bool VERTICAL>
__global__ __launch_bounds__(256) void k_conv(ConvParams P) {
	int x = blockIdx.x * blockDim.x + threadIdx.x;
	int y = blockIdx.y;
	const int extent = VERTICAL ? P.height : P.width;
	const int offL = P.koff, offR = P.kw - P.koff - 1;
	if (P.borderOnly) {
		// border fix-up after a streaming kernel: index t of the filtered axis runs over the offL leading and offR trailing positions
		// (horizontal: the grid is flat, consecutive threads take the border columns of one row, then the next row)
		const int nb = offL + offR;
		if (!VERTICAL) {
			if (nb <= 0) return;
			y = x / nb;
			x -= y * nb;
			if (y >= P.height) return;
		}
		int& t = VERTICAL ? y : x;
		if (t >= nb) return;
		t = t < offL ? t : extent - offR + (t - offL);
	}
	if (x >= P.width) return;
	const int pos = VERTICAL ? y : x;
	const long long step = VERTICAL ? P.inStride : 1;
	const float* src = P.in + (long long)blockIdx.z * P.inImageStride + (long long)y * P.inStride + x;
	const bool interior = pos >= offL && pos < extent - offR;
	float result;
	if (interior && P.mode == 3) return;
	if (interior && P.mode != 2) {
		const float* s = src - offL * step;
		result = P.unrolled ? tapsUnrolled(s, step, P.k, P.kw) : tapsStandard(s, step, P.k, P.kw);
	} else {
		if (P.mode == 0) return;
		const int k0 = max(0, offL - pos);
		const int k1 = min(P.kw, extent - pos + offL);
		float total = 0, weight = 0;
		for (int k = k0; k < k1; k++) {
			const float w = P.k[k];
			weight += w;
			total += src[(long long)(k - offL) * step] * w;
		}
		result = total / weight;
	}
	P.out[(long long)blockIdx.z * P.outImageStride + (long long)y * P.outStride + x] = result;
}

// ---- tiled forms: rows staged once in LDS with 16-byte loads, every tap served from LDS ----
// One output value from `s` (taps `step` floats apart in LDS), exactly the reference's expression for the pixel's position class:
//   interior            total = s0*k0 (unrolled widths) or 0 + s0*k0 (standard), then total += s_i*k_i in tap order
//   border, normalised  weight += k; total += s*k over the taps inside the image, total / weight   (ConvolveNormalized_JustBorder_SB)
// KW > 0: compile-time width (the reference's unrolled widths 3..11), KW == 0: run-time width.  Returns false when the pixel is not written.
// kc: where the coefficients are read from -- the kernel arguments by default; the tiled kernels pass their LDS copy (a run-time index into
// the argument block is a memory access per tap)
template <int KW>
__device__ __forceinline__ bool convOne(const ConvParams& P, const float* s, int step, int pos, int extent, float& result, const float* kc = nullptr) {
	if (!kc) kc = P.k;
	const int kw = KW > 0 ? KW : P.kw;
	const int offL = P.koff, offR = kw - P.koff - 1;
	const bool interior = pos >= offL && pos < extent - offR;
	if (interior && P.mode != 2) {
		if (P.mode == 3) return false;
		float total;
		if (KW > 0) {
			float v[KW > 0 ? KW : 1];
#pragma unroll
			for (int i = 0; i < KW; i++) v[i] = s[i * step];
			total = v[0] * kc[0];
#pragma unroll
			for (int i = 1; i < KW; i++) total += v[i] * kc[i];
		} else {
			total = P.unrolled ? s[0] * kc[0] : 0.0f + s[0] * kc[0];
			int i = 1;
			for (; i + 4 <= kw; i += 4) {
				const float v0 = s[i * step], v1 = s[(i + 1) * step], v2 = s[(i + 2) * step], v3 = s[(i + 3) * step];
				total += v0 * kc[i];
				total += v1 * kc[i + 1];
				total += v2 * kc[i + 2];
				total += v3 * kc[i + 3];
			}
			for (; i < kw; i++) total += s[i * step] * kc[i];
		}
		result = total;
		return true;
	}
	if (P.mode == 0) return false;
	const int k0 = max(0, offL - pos);
	const int k1 = min(kw, extent - pos + offL);
	float total = 0, weight = 0;
	for (int k = k0; k < k1; k++) {
		const float w = kc[k];
		weight += w;
		total += s[k * step] * w;
	}
	result = total / weight;
	return true;
}

// 16-byte load of in[gx .. gx+3] of a row of `width` floats; elements outside [0,width) read as 0 (they are never used as taps)
__device__ __forceinline__ float4 loadRow4(const float* __restrict__ row, int gx, int width) {
	if (gx >= 0 && gx + 3 < width) return *reinterpret_cast<const float4*>(row + gx);
	float4 v;
	v.x = (gx >= 0 && gx < width) ? row[gx] : 0.0f;
	v.y = (gx + 1 >= 0 && gx + 1 < width) ? row[gx + 1] : 0.0f;
	v.z = (gx + 2 >= 0 && gx + 2 < width) ? row[gx + 2] : 0.0f;
	v.w = (gx + 3 >= 0 && gx + 3 < width) ? row[gx + 3] : 0.0f;
	return v;
}

// Horizontal pass.  A block takes CT_ROWS rows x 256 columns; wave w stages and filters rows w, w+4, ...: the row segment plus the
// kernel's reach (rounded to 16-byte chunks) goes to LDS once, then lane l produces columns l, l+64, l+128, l+192 of the segment, so
// every LDS read and every global store of a wave touches 64 consecutive floats.
// (Measured alternative that lost: a wave walking a strip of 16 rows through two LDS row buffers of its own, the next row's loads under the
// tap loop, no block barrier, a quarter of the workgroups: 1.16 ms against 0.85 ms for 41 taps on 64 x 1080p.)
#define CT_W 256
#define CT_ROWS 8
template <int KW>
__global__ __launch_bounds__(256) void k_conv_h_tile(ConvParams P, int padL, int ldsRow) {
	extern __shared__ float lds[];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int xt0 = blockIdx.x * CT_W;
	const float* img = P.in + (long long)blockIdx.z * P.inImageStride;
	float* outImg = P.out + (long long)blockIdx.z * P.outImageStride;
	const int nChunks = ldsRow >> 2;
	const int y0 = blockIdx.y * CT_ROWS;
#pragma unroll
	for (int q = 0; q < CT_ROWS / 4; q++) {
		const int y = y0 + wave + 4 * q;
		if (y >= P.height) break;
		const float* src = img + (long long)y * P.inStride;
		float* row = lds + (wave + 4 * q) * ldsRow;
		for (int c = lane; c < nChunks; c += 64) *reinterpret_cast<float4*>(row + 4 * c) = loadRow4(src, xt0 - padL + 4 * c, P.width);
	}
	// the kernel's coefficients behind the rows: one broadcast LDS read per tap serves the lane's four outputs
	float* kl = lds + CT_ROWS * ldsRow;
	if ((int)threadIdx.x < P.kw) kl[threadIdx.x] = P.k[threadIdx.x];
	__syncthreads();
	const int kw = KW > 0 ? KW : P.kw;
	// Run-time widths: the reference's interior expression -- total = (0 +) s0*k0; total += s_i*k_i in tap order -- is formed for the four
	// outputs of a lane together (every tap of every output lies inside the staged row, whose cells outside the image hold 0 and are only
	// ever combined into values that are not stored); interior outputs are stored, the few border outputs of the frame tiles take convOne.
	const bool fast = KW == 0 && P.mode != 2 && P.mode != 3;
	const int offRh = kw - P.koff - 1;
#pragma unroll
	for (int q = 0; q < CT_ROWS / 4; q++) {
		const int y = y0 + wave + 4 * q;
		if (y >= P.height) break;
		const float* row = lds + (wave + 4 * q) * ldsRow + padL - P.koff;
		float* dst = outImg + (long long)y * P.outStride;
		if (fast) {
			// (the two taps of a pair are 64 words apart: one ds_read2st64_b32 fills a register pair for the packed multiply-add)
			typedef float f32x2 __attribute__((ext_vector_type(2)));
			const float* s0 = row + lane;
			const float k0 = kl[0];
			f32x2 a = f32x2{s0[0], s0[64]} * k0, b = f32x2{s0[128], s0[192]} * k0;
			if (!P.unrolled) { a = f32x2{0.0f, 0.0f} + a; b = f32x2{0.0f, 0.0f} + b; }
			// Four taps per step, the eight pair reads written out: left to itself the compiler pairs ADJACENT taps of one column (ds_read2_b32)
			// and then spends two moves per tap on re-pairing them for the packed arithmetic (910 vector instructions per wave instead of ~500).
			int i = 1;
			unsigned int ad = (unsigned int)(uintptr_t)(s0 + 1);   // LDS byte address of tap 1 of the lane's first output
			for (; i + 4 <= kw; i += 4, ad += 16u) {
				f32x2 p0, p1, p2, p3, q0, q1, q2, q3;
				const unsigned int a1 = ad + 4u, a2 = ad + 8u, a3 = ad + 12u;
				const float ka = kl[i], kb = kl[i + 1], kc = kl[i + 2], kd = kl[i + 3];   // issued first: the wait below covers them as well
				asm volatile("ds_read2st64_b32 %0, %8 offset0:0 offset1:1\n\t"
							 "ds_read2st64_b32 %4, %8 offset0:2 offset1:3\n\t"
							 "ds_read2st64_b32 %1, %9 offset0:0 offset1:1\n\t"
							 "ds_read2st64_b32 %5, %9 offset0:2 offset1:3\n\t"
							 "ds_read2st64_b32 %2, %10 offset0:0 offset1:1\n\t"
							 "ds_read2st64_b32 %6, %10 offset0:2 offset1:3\n\t"
							 "ds_read2st64_b32 %3, %11 offset0:0 offset1:1\n\t"
							 "ds_read2st64_b32 %7, %11 offset0:2 offset1:3\n\t"
							 "s_waitcnt lgkmcnt(0)"
							 : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3)
							 : "v"(ad), "v"(a1), "v"(a2), "v"(a3)
							 : "memory");
				a += p0 * ka; b += q0 * ka;
				a += p1 * kb; b += q1 * kb;
				a += p2 * kc; b += q2 * kc;
				a += p3 * kd; b += q3 * kd;
			}
			for (; i < kw; i++) {
				const float k = kl[i];
				a += f32x2{s0[i], s0[64 + i]} * k;
				b += f32x2{s0[128 + i], s0[192 + i]} * k;
			}
			const float t[4] = {a.x, a.y, b.x, b.y};
#pragma unroll
			for (int j = 0; j < 4; j++) {
				const int xl = lane + 64 * j, x = xt0 + xl;
				if (x >= P.width) continue;
				if (x >= P.koff && x < P.width - offRh) dst[x] = t[j];
				else {
					float r;
					if (convOne<KW>(P, row + xl, 1, x, P.width, r, kl)) dst[x] = r;
				}
			}
			continue;
		}
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const int xl = lane + 64 * j, x = xt0 + xl;
			float r;
			if (x < P.width && convOne<KW>(P, row + xl, 1, x, P.width, r, kl)) dst[x] = r;
		}
	}
}

// Vertical pass.  A block takes CV_ROWS output rows x 256 columns: the CV_ROWS + kw - 1 input rows are staged in LDS (one row per wave
// instruction, 16 bytes per lane), then wave w filters rows w, w+4, ...; lane l owns columns 4l .. 4l+3, so taps are ds_read_b128
// and results leave as 16-byte stores.
#define CV_ROWS 32
// NW = waves per block.  The run-time widths (wide kernels) run 16 waves per block: a block's LDS is (CV_ROWS + kw - 1) KB -- 72 KB for 41 taps, two
// blocks per CU -- so four waves per block left two waves per SIMD to hide the LDS latency of a tap loop (41 taps: 0.81 -> 0.53 ms per 64 x 1080p).
template <int KW, int NW = 4>
__global__ __launch_bounds__(64 * NW) void k_conv_v_tile(ConvParams P) {
	extern __shared__ float lds[];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int x = blockIdx.x * CT_W + 4 * lane;
	const int y0 = blockIdx.y * CV_ROWS;
	const float* img = P.in + (long long)blockIdx.z * P.inImageStride;
	float* outImg = P.out + (long long)blockIdx.z * P.outImageStride;
	const int kw = KW > 0 ? KW : P.kw;
	const int nStage = min(CV_ROWS, P.height - y0) + kw - 1;
	const int yTop = y0 - P.koff;
	if (x < P.width) {
		for (int i = wave; i < nStage; i += NW) {
			const int yy = yTop + i;
			if (yy >= 0 && yy < P.height) *reinterpret_cast<float4*>(lds + i * CT_W + 4 * lane) = loadRow4(img + (long long)yy * P.inStride, x, P.width);
		}
	}
	float* kl = lds + (CV_ROWS + kw - 1) * CT_W;   // the kernel's coefficients behind the rows (one broadcast read per tap)
	if ((int)threadIdx.x < kw) kl[threadIdx.x] = P.k[threadIdx.x];
	__syncthreads();
	if (x >= P.width) return;
	const int offR = kw - P.koff - 1;
	for (int q = wave; q < CV_ROWS; q += NW) {
		const int y = y0 + q;
		if (y >= P.height) break;
		const float* s = lds + q * CT_W + 4 * lane;
		if (KW == 0 && P.mode != 2 && P.mode != 3 && y >= P.koff && y < P.height - offR && x + 3 < P.width) {
			// interior row (wave-uniform): the lane's four columns together, taps as 16-byte LDS reads, packed fp32 -- per column the
			// reference's expression, total = (0 +) s0*k0; total += s_i*k_i in tap order
			typedef float f32x2 __attribute__((ext_vector_type(2)));
			const float4 v0 = *reinterpret_cast<const float4*>(s);
			const float k0 = kl[0];
			f32x2 lo = f32x2{v0.x, v0.y} * k0, hi = f32x2{v0.z, v0.w} * k0;
			if (!P.unrolled) { lo = f32x2{0.0f, 0.0f} + lo; hi = f32x2{0.0f, 0.0f} + hi; }
#pragma unroll 4
			for (int i = 1; i < kw; i++) {
				const float4 v = *reinterpret_cast<const float4*>(s + i * CT_W);
				const float k = kl[i];
				lo += f32x2{v.x, v.y} * k;
				hi += f32x2{v.z, v.w} * k;
			}
			*reinterpret_cast<float4*>(outImg + (long long)y * P.outStride + x) = make_float4(lo.x, lo.y, hi.x, hi.y);
			continue;
		}
		float r[4];
		bool wr = true;
#pragma unroll
		for (int j = 0; j < 4; j++) wr = convOne<KW>(P, s + j, CT_W, y, P.height, r[j], kl);   // same position class for the four columns
		if (!wr) continue;
		float* dst = outImg + (long long)y * P.outStride + x;
		if (x + 3 < P.width) *reinterpret_cast<float4*>(dst) = make_float4(r[0], r[1], r[2], r[3]);
		else
			for (int j = 0; j < 4 && x + j < P.width; j++) dst[j] = r[j];
	}
}

// ---- streaming forms for the reference's unrolled widths (3..11, centred): no LDS, every input row loaded once per strip ----
// Horizontal: lane l owns columns 4l..4l+3 of a 256-column strip and walks CS_ROWS rows; a row's taps are NL aligned 16-byte loads
// (own chunk + the chunks the kernel reaches into; neighbouring lanes' chunks are L1 hits), results leave as one 16-byte store.
#define CS_ROWS 8
template <int KW>
__global__ __launch_bounds__(256) void k_conv_h_stream(ConvParams P) {
	constexpr int R = KW / 2, NC = (R + 3) / 4, NL = 1 + 2 * NC, PL = 4 * NC;
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int x = blockIdx.x * 256 + 4 * lane;
	const int y0 = (blockIdx.y * 4 + wave) * CS_ROWS;
	if (x >= P.width || y0 >= P.height) return;
	const float* img = P.in + (long long)blockIdx.z * P.inImageStride;
	float* outImg = P.out + (long long)blockIdx.z * P.outImageStride;
	const int yEnd = min(y0 + CS_ROWS, P.height);
	// interior pixels only: the border columns (normalised forms) are written afterwards by the general kernel (borderOnly launch)
	const bool allInterior = x >= R && x + 3 < P.width - R;
	// NB row buffers take turns: while row y is filtered and stored, the chunks of the next NB - 1 rows are in flight (a wave has to keep
	// several KB on the way to cover the HBM latency at 8 waves per SIMD)
	constexpr int NB = KW <= 5 ? 4 : (KW <= 7 ? 3 : 2);
	if constexpr (NC == 1) {
		// kernels that reach at most one chunk to either side: every lane loads its own chunk only and receives the neighbours' chunks by
		// lane shifts; the strip's first / last lane fetch the chunk beyond the strip themselves (one extra 16-byte request per row and side)
		float4 own[NB], edge[NB];
		const bool first = lane == 0, last = lane == 63;
		auto fetch = [&](float4& own, float4& edge, int y) {
			if (y < yEnd) {
				const float* row = img + (long long)y * P.inStride;
				own = loadRow4(row, x, P.width);
				if (first) edge = loadRow4(row, x - 4, P.width);
				if (last) edge = loadRow4(row, x + 4, P.width);
			}
		};
		auto emit = [&](const float4& own, const float4& edge, int y) {
			if (y >= yEnd) return;   // wave-uniform
			float4 lf, rt;
			lf.x = __shfl_up(own.x, 1, 64); lf.y = __shfl_up(own.y, 1, 64); lf.z = __shfl_up(own.z, 1, 64); lf.w = __shfl_up(own.w, 1, 64);
			rt.x = __shfl_down(own.x, 1, 64); rt.y = __shfl_down(own.y, 1, 64); rt.z = __shfl_down(own.z, 1, 64); rt.w = __shfl_down(own.w, 1, 64);
			if (first) lf = edge;
			if (last) rt = edge;
			const float v[12] = {lf.x, lf.y, lf.z, lf.w, own.x, own.y, own.z, own.w, rt.x, rt.y, rt.z, rt.w};
			float r[4];
#pragma unroll
			for (int j = 0; j < 4; j++) {
				float total = v[PL - R + j] * P.k[0];
#pragma unroll
				for (int i = 1; i < KW; i++) total += v[PL - R + j + i] * P.k[i];
				r[j] = total;
			}
			float* dst = outImg + (long long)y * P.outStride + x;
			if (allInterior) *reinterpret_cast<float4*>(dst) = make_float4(r[0], r[1], r[2], r[3]);
			else {
#pragma unroll
				for (int j = 0; j < 4; j++)
					if (x + j >= R && x + j < P.width - R) dst[j] = r[j];
			}
		};
#pragma unroll
		for (int q = 0; q < NB; q++) fetch(own[q], edge[q], y0 + q);
		for (int y = y0; y < yEnd; y += NB) {
#pragma unroll
			for (int q = 0; q < NB; q++) {
				emit(own[q], edge[q], y + q);
				fetch(own[q], edge[q], y + q + NB);
			}
		}
		return;
	}
	float4 buf[NB][NL];
	auto fetch = [&](float4 (&buf)[NL], int y) {
		if (y < yEnd) {
			const float* row = img + (long long)y * P.inStride;
#pragma unroll
			for (int c = 0; c < NL; c++) buf[c] = loadRow4(row, x - PL + 4 * c, P.width);
		}
	};
	auto emit = [&](const float4 (&buf)[NL], int y) {
		if (y >= yEnd) return;
		float v[4 * NL];
#pragma unroll
		for (int c = 0; c < NL; c++) { v[4 * c] = buf[c].x; v[4 * c + 1] = buf[c].y; v[4 * c + 2] = buf[c].z; v[4 * c + 3] = buf[c].w; }
		float r[4];
#pragma unroll
		for (int j = 0; j < 4; j++) {
			float total = v[PL - R + j] * P.k[0];
#pragma unroll
			for (int i = 1; i < KW; i++) total += v[PL - R + j + i] * P.k[i];
			r[j] = total;
		}
		float* dst = outImg + (long long)y * P.outStride + x;
		if (allInterior) *reinterpret_cast<float4*>(dst) = make_float4(r[0], r[1], r[2], r[3]);
		else {
#pragma unroll
			for (int j = 0; j < 4; j++)
				if (x + j >= R && x + j < P.width - R) dst[j] = r[j];
		}
	};
#pragma unroll
	for (int q = 0; q < NB; q++) fetch(buf[q], y0 + q);
	for (int y = y0; y < yEnd; y += NB) {
#pragma unroll
		for (int q = 0; q < NB; q++) {
			emit(buf[q], y + q);
			fetch(buf[q], y + q + NB);
		}
	}
}

// Vertical: lane l owns columns 4l..4l+3 and walks a strip of CS_ROWS_V output rows with the KW input rows of the current output in a
// register ring of KW+PF slots (the extra slots receive the rows of the next PF outputs while the current one is computed).  Strips that touch the top or
// bottom border evaluate the border rules per row from the same ring.
#define CS_ROWS_V 16
template <int KW>
__global__ __launch_bounds__(256) void k_conv_v_stream(ConvParams P) {
	constexpr int PF = KW <= 5 ? 3 : 2;        // rows requested ahead of the output being computed
	constexpr int R = KW / 2, RING = KW + PF;
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int x = blockIdx.x * 256 + 4 * lane;
	const int y0 = (blockIdx.y * 4 + wave) * CS_ROWS_V;
	if (x >= P.width || y0 >= P.height) return;
	const float* img = P.in + (long long)blockIdx.z * P.inImageStride + x;
	float* outImg = P.out + (long long)blockIdx.z * P.outImageStride + x;
	const int yEnd = min(y0 + CS_ROWS_V, P.height);
	const bool full4 = x + 3 < P.width;
	auto loadRow = [&](int yy) -> float4 {
		if (yy < 0 || yy >= P.height) return make_float4(0, 0, 0, 0);
		return loadRow4(img + (long long)yy * P.inStride - x, x, P.width);
	};
	float4 ring[RING];
	// ring[(t + i) % RING] = input row y0 - R + t + i, the tap i of output row y0 + t
#pragma unroll
	for (int i = 0; i < KW + PF - 1; i++) ring[i] = loadRow(y0 - R + i);
	for (int tb = 0; y0 + tb < yEnd; tb += RING) {
#pragma unroll
		for (int j = 0; j < RING; j++) {
			const int y = y0 + tb + j;
			if (y < yEnd) {
				if (y + PF < yEnd) ring[(j + KW + PF - 1) % RING] = loadRow(y + PF + R);   // a spare slot receives the last tap of the output PF rows on
				float r[4];
				const float4 t0 = ring[j % RING];
				r[0] = t0.x * P.k[0]; r[1] = t0.y * P.k[0]; r[2] = t0.z * P.k[0]; r[3] = t0.w * P.k[0];
#pragma unroll
				for (int i = 1; i < KW; i++) {
					const float4 t = ring[(j + i) % RING];
					r[0] += t.x * P.k[i]; r[1] += t.y * P.k[i]; r[2] += t.z * P.k[i]; r[3] += t.w * P.k[i];
				}
				if (y >= R && y < P.height - R) {   // interior rows only: the border rows are written by the general kernel (borderOnly launch)
					float* dst = outImg + (long long)y * P.outStride;
					if (full4) *reinterpret_cast<float4*>(dst) = make_float4(r[0], r[1], r[2], r[3]);
					else
						for (int q = 0; q < 4 && x + q < P.width; q++) dst[q] = r[q];
				}
			}
		}
	}
}

// ---- Gaussian blur in ONE pass over the image (BlurImageOps.gaussian = ConvolveImageNormalized.horizontal into `storage`, then
// ConvolveImageNormalized.vertical into the output: I:alg/filter/blur/BlurImageOps.java:406-425) for the reference's unrolled widths.
// A wave owns a strip of 256 columns x BF_ROWS output rows and walks down the input rows once: a row is loaded (16 bytes per lane),
// filtered horizontally in registers -- the value the reference would have stored in `storage`, border columns included -- and pushed
// into a register ring of the last KW filtered rows, from which the vertical filter produces one output row per input row.  The
// intermediate image never exists: 4P read + 4P written (+ 2R re-read rows per strip) instead of 16P for the two passes.  Every value is
// formed by the reference's expression for its position class, in its tap order:
//   interior  total = s0*k0; total += s_i*k_i                                   (ConvolveImageUnrolled_SB_F32_F32.java:152-180,384-425)
//   border    weight += k; total += s*k over the taps inside the image; total / weight   (ConvolveNormalized_JustBorder_SB.java:60-88,108-140)
#define BF_ROWS 32   // output rows per wave strip
// Measured alternatives that lost (A/B in one run, 64 x 1080p, r = 2 / r = 5: this form 0.27 / 0.64 ms): chunk-bounds tests hoisted out of
// the row loop with plain 16-byte loads 0.31 / 0.87 ms; one row of look-ahead instead of two for widths 9, 11 (107 instead of 126
// registers) 0.76 ms; 64-row strips for widths 9, 11 0.65 ms at 1080p but 0.31 instead of 0.25 ms on 8 x 4K.
template <int KW>
__device__ __forceinline__ float blurTapsInterior(const float (&v)[KW], const float* k) {
	float total = v[0] * k[0];
#pragma unroll
	for (int i = 1; i < KW; i++) total += v[i] * k[i];
	return total;
}
// first: index of tap 0 along the filtered axis (pos - R); extent: image size along that axis.  Taps outside [0, extent) are skipped.
template <int KW>
__device__ __forceinline__ float blurTapsBorder(const float (&v)[KW], const float* k, int first, int extent) {
	float total = 0, weight = 0;
#pragma unroll
	for (int i = 0; i < KW; i++) {
		const bool in = first + i >= 0 && first + i < extent;
		const float w = k[i];
		weight = in ? weight + w : weight;
		total = in ? total + v[i] * w : total;
	}
	return total / weight;
}
template <int KW, int WPB = 4>
__global__ __launch_bounds__(64 * WPB) void k_blur_fused(ConvParams P) {
	constexpr int R = KW / 2, NC = (R + 3) / 4, NL = 1 + 2 * NC, PL = 4 * NC;
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int x = blockIdx.x * 256 + 4 * lane;
	constexpr int ROWS = BF_ROWS;
	const int y0 = (blockIdx.y * WPB + wave) * ROWS;
	if (x >= P.width || y0 >= P.height) return;
	const int W = P.width, H = P.height;
	const float* img = P.in + (long long)blockIdx.z * P.inImageStride;
	float* outImg = P.out + (long long)blockIdx.z * P.outImageStride + x;
	const int yEnd = min(y0 + ROWS, H);
	const bool colsInterior = x >= R && x + 3 < W - R;     // all four columns are interior columns of the horizontal pass
	const bool full4 = x + 3 < W;
	const int yLast = min(yEnd + R, H);                    // input rows of the strip: [max(y0 - R, 0), yLast)
	// the lane's chunk of a row and the chunks the kernel reaches into (neighbouring lanes' chunks: L1 hits)
	auto fetch = [&](float4 (&buf)[NL], int yy) {
		if (yy >= 0 && yy < yLast) {
			const float* row = img + (long long)yy * P.inStride;
#pragma unroll
			for (int c = 0; c < NL; c++) buf[c] = loadRow4(row, x - PL + 4 * c, W);
		}
	};
	// horizontal pass of one row for the lane's four columns: the value ConvolveImageNormalized.horizontal leaves in `storage`
	auto hfilter = [&](const float4 (&buf)[NL]) -> float4 {
		float v[4 * NL];
#pragma unroll
		for (int c = 0; c < NL; c++) { v[4 * c] = buf[c].x; v[4 * c + 1] = buf[c].y; v[4 * c + 2] = buf[c].z; v[4 * c + 3] = buf[c].w; }
		if (colsInterior) {
			// the four columns as two packed pairs: column j's tap i is v[PL - R + j + i], so the pairs are neighbouring array elements;
			// component-wise the reference's expression (total = s0*k0; total += s_i*k_i), half the additions of four scalar chains
			typedef float f32x2 __attribute__((ext_vector_type(2)));
			f32x2 lo = f32x2{v[PL - R], v[PL - R + 1]} * P.k[0], hi = f32x2{v[PL - R + 2], v[PL - R + 3]} * P.k[0];
#pragma unroll
			for (int i = 1; i < KW; i++) {
				lo += f32x2{v[PL - R + i], v[PL - R + 1 + i]} * P.k[i];
				hi += f32x2{v[PL - R + 2 + i], v[PL - R + 3 + i]} * P.k[i];
			}
			return make_float4(lo.x, lo.y, hi.x, hi.y);
		}
		float r[4];
#pragma unroll
		for (int j = 0; j < 4; j++) {
			float t[KW];
#pragma unroll
			for (int i = 0; i < KW; i++) t[i] = v[PL - R + j + i];
			r[j] = (x + j >= R && x + j < W - R) ? blurTapsInterior<KW>(t, P.k) : blurTapsBorder<KW>(t, P.k, x + j - R, W);
		}
		return make_float4(r[0], r[1], r[2], r[3]);
	};
	// ring[i] = horizontally filtered row (yy - 2R + i) once input row yy has been pushed: the KW taps of output row yy - R, in tap order.
	// The ring is shifted by register moves (KW x 4 per row, against 16 KW multiply-adds), which keeps the row loop one small body.
	float4 ring[KW];
#pragma unroll
	for (int i = 0; i < KW; i++) ring[i] = make_float4(0, 0, 0, 0);
	float4 cur[NL], nxt[NL];   // two rows in flight ahead of the one being filtered
#pragma unroll
	for (int c = 0; c < NL; c++) { cur[c] = make_float4(0, 0, 0, 0); nxt[c] = make_float4(0, 0, 0, 0); }
	fetch(cur, y0 - R);
	fetch(nxt, y0 - R + 1);
#pragma unroll 1
	for (int yy = y0 - R; yy < yEnd + R; yy++) {
		float4 hrow = make_float4(0, 0, 0, 0);
		if (yy >= 0 && yy < H) hrow = hfilter(cur);   // rows outside the image are never used as taps (wave-uniform branch)
#pragma unroll
		for (int c = 0; c < NL; c++) cur[c] = nxt[c];
		fetch(nxt, yy + 2);
#pragma unroll
		for (int i = 0; i + 1 < KW; i++) ring[i] = ring[i + 1];
		ring[KW - 1] = hrow;
		const int y = yy - R;   // the output row whose last tap just arrived
		if (y >= y0) {
			const bool rowInterior = y >= R && y < H - R;   // wave-uniform
			float r[4];
			if (rowInterior) {
				// the four columns as two packed pairs (v_pk_mul_f32 / v_pk_add_f32): component-wise the reference's expression, half the instructions
				typedef float f32x2 __attribute__((ext_vector_type(2)));
				f32x2 lo = f32x2{ring[0].x, ring[0].y} * P.k[0], hi = f32x2{ring[0].z, ring[0].w} * P.k[0];
#pragma unroll
				for (int i = 1; i < KW; i++) {
					lo += f32x2{ring[i].x, ring[i].y} * P.k[i];
					hi += f32x2{ring[i].z, ring[i].w} * P.k[i];
				}
				r[0] = lo.x; r[1] = lo.y; r[2] = hi.x; r[3] = hi.y;
			} else {
#pragma unroll
				for (int q = 0; q < 4; q++) {
					float tv[KW];
#pragma unroll
					for (int i = 0; i < KW; i++) tv[i] = q == 0 ? ring[i].x : q == 1 ? ring[i].y : q == 2 ? ring[i].z : ring[i].w;
					r[q] = blurTapsBorder<KW>(tv, P.k, y - R, H);
				}
			}
			float* dst = outImg + (long long)y * P.outStride;
			if (full4) *reinterpret_cast<float4*>(dst) = make_float4(r[0], r[1], r[2], r[3]);
			else
				for (int q = 0; q < 4 && x + q < W; q++) dst[q] = r[q];
		}
	}
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int bhip_launch_conv(bhip_ctx* ctx, bool vertical, bool normalized, const float* kernel, int kw, int koff, const float* in, int inStride, int width,
					 int height, float* out, int outStride, int batch, long long inImageStride, long long outImageStride) {
	if (kw <= 0 || kw > BHIP_MAX_TAPS || koff < 0 || koff >= kw) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "kernel width not supported");
	if (width <= 0 || height <= 0 || batch <= 0) return BHIP_OK;
	ConvParams P;
	P.in = in; P.out = out; P.inStride = inStride; P.outStride = outStride; P.width = width; P.height = height; P.kw = kw; P.koff = koff;
	P.inImageStride = inImageStride; P.outImageStride = outImageStride;
	for (int i = 0; i < kw; i++) P.k[i] = kernel[i];
	P.unrolled = (koff == kw / 2 && kw % 2 == 1 && (kw == 3 || kw == 5 || kw == 7 || kw == 9 || kw == 11)) ? 1 : 0;
	P.mode = 0;
	P.borderOnly = 0;
	if (normalized) {
		const int extent = vertical ? height : width;
		if (kw >= extent) {
			P.mode = 2;
		} else {
			P.mode = 1;
			// ConvolveImageNormalized: re-normalise when |sum - 1| > 1e-4 (Kernel1D_F32.computeSum is a sequential fp32 sum)
			float sum = 0;
			for (int i = 0; i < kw; i++) sum += P.k[i];
			float diff = sum - 1.0f;
			if (diff < 0) diff = -diff;
			if (diff > 1e-4f) {
				float total = 0;
				for (int i = 0; i < kw; i++) total += P.k[i];
				for (int i = 0; i < kw; i++) P.k[i] /= total;
			}
		}
	}
	ProfScope prof(ctx, vertical ? "k_conv_v" : "k_conv_h", 8.0 * width * height * batch);
	// tiled forms need 16-byte aligned rows on both sides; the naive form (kernel wider than the image) stays on the general kernel
	const bool tiled = P.mode != 2 && aligned16(in) && aligned16(out) && inStride % 4 == 0 && outStride % 4 == 0 && inImageStride % 4 == 0 &&
					   outImageStride % 4 == 0 && kw <= 97;
	if (!tiled) {
		dim3 grid((width + 255) / 256, height, batch);
		if (vertical) hipLaunchKernelGGL(k_conv<true>, grid, dim3(256), 0, ctx->stream, P);
		else hipLaunchKernelGGL(k_conv<false>, grid, dim3(256), 0, ctx->stream, P);
	} else if (P.unrolled) {
		// the reference's unrolled widths: register-streaming kernels
		const int rows = vertical ? CS_ROWS_V : CS_ROWS;
		dim3 grid((width + 255) / 256, (height + 4 * rows - 1) / (4 * rows), batch);
#define LAUNCH_S(KWT)                                                                                      \
	do {                                                                                                   \
		if (vertical) hipLaunchKernelGGL(k_conv_v_stream<KWT>, grid, dim3(256), 0, ctx->stream, P);        \
		else hipLaunchKernelGGL(k_conv_h_stream<KWT>, grid, dim3(256), 0, ctx->stream, P);                 \
	} while (0)
		switch (kw) {
		case 3: LAUNCH_S(3); break;
		case 5: LAUNCH_S(5); break;
		case 7: LAUNCH_S(7); break;
		case 9: LAUNCH_S(9); break;
		default: LAUNCH_S(11); break;
		}
#undef LAUNCH_S
		if (P.mode == 1) {
			// the kw - 1 border columns (rows) with the re-normalised formula: a thin launch of the general kernel
			P.borderOnly = 1;
			if (vertical) hipLaunchKernelGGL(k_conv<true>, dim3((width + 255) / 256, kw - 1, batch), dim3(256), 0, ctx->stream, P);
			else hipLaunchKernelGGL(k_conv<false>, dim3((unsigned)(((long long)height * (kw - 1) + 255) / 256), 1, batch), dim3(256), 0, ctx->stream, P);
		}
	} else if (vertical) {
		dim3 grid((width + CT_W - 1) / CT_W, (height + CV_ROWS - 1) / CV_ROWS, batch);
		const size_t ldsBytes = (size_t)(CV_ROWS + kw - 1) * CT_W * 4 + (size_t)kw * 4;   // rows + the kernel
		if (ldsBytes > 65536) BHIP_HIP(ctx, hipFuncSetAttribute((const void*)k_conv_v_tile<0, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
		hipLaunchKernelGGL((k_conv_v_tile<0, 16>), grid, dim3(1024), ldsBytes, ctx->stream, P);
	} else {
		const int offR = kw - koff - 1;
		const int padL = (koff + 3) & ~3, padR = (offR + 3) & ~3;
		const int ldsRow = CT_W + padL + padR;
		dim3 grid((width + CT_W - 1) / CT_W, (height + CT_ROWS - 1) / CT_ROWS, batch);
		const size_t ldsBytes = (size_t)CT_ROWS * ldsRow * 4 + (size_t)kw * 4;   // rows + the kernel
#define LAUNCH_H(KWT) hipLaunchKernelGGL(k_conv_h_tile<KWT>, grid, dim3(256), ldsBytes, ctx->stream, P, padL, ldsRow)
		LAUNCH_H(0);
#undef LAUNCH_H
	}
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}

// BlurImageOps.gaussian in one pass; returns *done = false when the shape / kernel is outside what the fused kernel covers (the caller
// then runs the two separable passes)
int bhip_launch_blur_fused(bhip_ctx* ctx, const float* kernel, int kw, const float* in, int inStride, int width, int height, float* out, int outStride, int batch,
						   long long inImageStride, long long outImageStride, bool* done) {
	*done = false;
	if (width <= 0 || height <= 0 || batch <= 0) { *done = true; return BHIP_OK; }
	const bool unrolled = kw == 3 || kw == 5 || kw == 7 || kw == 9 || kw == 11;
	if (!unrolled || kw >= width || kw >= height) return BHIP_OK;   // standard-form widths and the naive form (kernel wider than the image)
	if (!(aligned16(in) && aligned16(out) && inStride % 4 == 0 && outStride % 4 == 0 && inImageStride % 4 == 0 && outImageStride % 4 == 0)) return BHIP_OK;
	if (bhip_env_flag("BHIP_BLUR_TWO_PASS")) return BHIP_OK;        // parity cross-check of the two forms
	ConvParams P;
	P.in = in; P.out = out; P.inStride = inStride; P.outStride = outStride; P.width = width; P.height = height; P.kw = kw; P.koff = kw / 2;
	P.inImageStride = inImageStride; P.outImageStride = outImageStride;
	for (int i = 0; i < kw; i++) P.k[i] = kernel[i];
	P.unrolled = 1; P.mode = 1; P.borderOnly = 0;
	{
		// ConvolveImageNormalized: re-normalise when |sum - 1| > 1e-4 (both passes see the same kernel)
		float sum = 0;
		for (int i = 0; i < kw; i++) sum += P.k[i];
		float diff = sum - 1.0f;
		if (diff < 0) diff = -diff;
		if (diff > 1e-4f) {
			float total = 0;
			for (int i = 0; i < kw; i++) total += P.k[i];
			for (int i = 0; i < kw; i++) P.k[i] /= total;
		}
	}
	// algorithmic bytes: the two separable passes of SURVEY 8d (8P each)
	ProfScope prof(ctx, "k_blur_fused", 16.0 * width * height * batch);
	const int rowsPerWave = BF_ROWS;
	dim3 grid((width + 255) / 256, (height + 4 * rowsPerWave - 1) / (4 * rowsPerWave), batch);
	switch (kw) {
	case 3: hipLaunchKernelGGL(k_blur_fused<3>, grid, dim3(256), 0, ctx->stream, P); break;
	case 5: hipLaunchKernelGGL(k_blur_fused<5>, grid, dim3(256), 0, ctx->stream, P); break;
	case 7: hipLaunchKernelGGL(k_blur_fused<7>, grid, dim3(256), 0, ctx->stream, P); break;
	// the wide kernels run one wave per workgroup: at 125 registers a CU holds 16 waves either way, but a wave slot is then free again as soon
	// as ITS strip is done instead of when the slowest of four is (r = 5 on 64 x 1080p: 0.64 -> 0.62 ms)
	case 9: hipLaunchKernelGGL((k_blur_fused<9, 1>), dim3(grid.x, (height + rowsPerWave - 1) / rowsPerWave, batch), dim3(64), 0, ctx->stream, P); break;
	default: hipLaunchKernelGGL((k_blur_fused<11, 1>), dim3(grid.x, (height + rowsPerWave - 1) / rowsPerWave, batch), dim3(64), 0, ctx->stream, P); break;
	}
	BHIP_HIP(ctx, hipGetLastError());
	*done = true;
	return BHIP_OK;
}

// ---------------- down-sampling convolution (pyramid layer step) ----------------
// ConvolveImageDownNormalized.horizontal/vertical   I:alg/filter/convolve/ConvolveImageDownNormalized.java:53-86
//   interior  I:alg/filter/convolve/down/ConvolveDownNoBorderUnrolled_F32_F32.java:140-154,351-366 (widths 3..11, first tap assigns)
//             I:alg/filter/convolve/down/ConvolveDownNoBorderStandard.java:43-110 (total = 0 first)
//   border    I:alg/filter/convolve/down/ConvolveDownNormalized_JustBorder.java:43-139
//   naive     I:alg/filter/convolve/down/ConvolveDownNormalizedNaive.java:40-94 (kernel.width >= image.width)
//   ranges    I:alg/filter/convolve/down/UtilDownConvolve.java:27-44
// The reference runs interior, then left border, then right border; a later loop overwrites an earlier one.  Per output index D
// along the filtered axis that order collapses to: right border if D*skip in [offsetEnd, sideTrunc); else left border if
// D*skip < offset; else interior centred on D*skip + offset%skip while that is <= maxSide; else the pixel is not written.
// (offset%skip != 0 only for skip >= 3 with radius > skip: the reference then samples off-grid and leaves one column untouched;
// reproduced as is.)
struct ConvDownParams {
	const float* in;
	float* out;
	long long inImageStride, outImageStride;
	int inStride, outStride, width, height;   // input size
	int skip, kw, radius;
	int unrolled, naive;
	int offset, offsetRem, maxSide, offsetEnd, sideTrunc;   // along the filtered axis
	int skipInterior;   // general kernel only: the interior outputs have been written by a streaming kernel
	float k[BHIP_MAX_TAPS];
};

template <bool VERTICAL>
__global__ __launch_bounds__(256) void k_conv_down(ConvDownParams P) {
	int ox = blockIdx.x * blockDim.x + threadIdx.x;
	int oy = blockIdx.y;
	const int outW = VERTICAL ? P.width : P.width / P.skip;
	if (P.skipInterior) {
		// border fix-up after a streaming kernel: the grid covers only the outputs whose centre lies left of `offset` or at / beyond
		// `offsetEnd` (horizontal: flat grid, consecutive threads take the border columns of one row, then the next row)
		const int outSide = (VERTICAL ? P.height : P.width) / P.skip;
		const int nl = min((P.offset + P.skip - 1) / P.skip, outSide), dR0 = max(nl, min((P.offsetEnd + P.skip - 1) / P.skip, outSide));
		const int nb = nl + (outSide - dR0);
		if (nb <= 0) return;
		int t;
		if (VERTICAL) { t = oy; }
		else { oy = ox / nb; t = ox - oy * nb; if (oy >= P.height) return; }
		if (t >= nb) return;
		const int D = t < nl ? t : dR0 + (t - nl);
		if (VERTICAL) oy = D; else ox = D;
	}
	if (ox >= outW) return;
	const float* img = P.in + (long long)blockIdx.z * P.inImageStride;
	const int D = VERTICAL ? oy : ox;
	const int side = VERTICAL ? P.height : P.width;
	const long long step = VERTICAL ? P.inStride : 1;
	const int r = P.radius;
	int centre = D * P.skip;
	int k0, k1;   // tap range [k0,k1] relative to the centre
	bool normalise = true;
	if (P.naive) {
		k0 = max(-r, -centre);
		k1 = min(r, side - 1 - centre);
	} else if (centre >= P.offsetEnd && centre < P.sideTrunc) {
		k0 = -r;
		k1 = min(r, side - centre - 1);
	} else if (centre < P.offset) {
		k0 = -centre;
		k1 = r;
	} else {
		centre += P.offsetRem;
		if (centre > P.maxSide || P.skipInterior) return;
		k0 = -r; k1 = r;
		normalise = false;
	}
	const float* src = VERTICAL ? img + (long long)centre * P.inStride + ox : img + (long long)oy * P.inStride + centre;
	float result;
	if (normalise) {
		float total = 0, weight = 0;
		for (int k = k0; k <= k1; k++) {
			const float w = P.k[k + r];
			weight += w;
			total += src[k * step] * w;
		}
		result = total / weight;
	} else {
		const float* s0 = src - r * step;
		result = P.unrolled ? tapsUnrolled(s0, step, P.k, P.kw) : tapsStandard(s0, step, P.k, P.kw);
	}
	P.out[(long long)blockIdx.z * P.outImageStride + (long long)oy * P.outStride + ox] = result;
}

// ---- streaming forms for skip 2 and the unrolled widths 3 and 5 (the discrete pyramid's usual layer step) ----
// Only INTERIOR outputs (the closed form above) are written here; the general kernel adds the border outputs afterwards (skipInterior).
__device__ __forceinline__ bool downInterior(const ConvDownParams& P, int D) {
	const int centre = D * P.skip;
	if (centre >= P.offsetEnd && centre < P.sideTrunc) return false;   // right border
	if (centre < P.offset) return false;                               // left border
	return centre + P.offsetRem <= P.maxSide;
}
// Horizontal: lane l owns input columns 4l..4l+3 of a 256-column strip = outputs 2l, 2l+1 (centres 4l and 4l+2; the kernel reaches at most
// two columns into the neighbouring chunks, which arrive by lane shifts); rows as in k_conv_h_stream.
#define CD_ROWS 8
template <int KW>
__global__ __launch_bounds__(256) void k_conv_down_h_stream(ConvDownParams P) {
	constexpr int R = KW / 2, NB = 4;
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int x = blockIdx.x * 256 + 4 * lane;          // first input column of this lane
	const int y0 = (blockIdx.y * 4 + wave) * CD_ROWS;
	if (x >= P.width || y0 >= P.height) return;
	const float* img = P.in + (long long)blockIdx.z * P.inImageStride;
	float* outImg = P.out + (long long)blockIdx.z * P.outImageStride;
	const int yEnd = min(y0 + CD_ROWS, P.height);
	const int outW = P.width / 2;
	const int D0 = x >> 1;                              // outputs D0, D0 + 1
	const bool w0 = D0 < outW && downInterior(P, D0), w1 = D0 + 1 < outW && downInterior(P, D0 + 1);
	const bool first = lane == 0, last = lane == 63;
	float4 own[NB], edge[NB];
	auto fetch = [&](float4& own, float4& edge, int y) {
		if (y < yEnd) {
			const float* row = img + (long long)y * P.inStride;
			own = loadRow4(row, x, P.width);
			if (first) edge = loadRow4(row, x - 4, P.width);
			if (last) edge = loadRow4(row, x + 4, P.width);
		}
	};
	auto emit = [&](const float4& own, const float4& edge, int y) {
		if (y >= yEnd) return;   // wave-uniform
		float4 lf, rt;
		lf.z = __shfl_up(own.z, 1, 64); lf.w = __shfl_up(own.w, 1, 64);
		rt.x = __shfl_down(own.x, 1, 64);
		if (first) { lf.z = edge.z; lf.w = edge.w; }
		if (last) rt.x = edge.x;
		// v[i] = input column x - 2 + i
		const float v[7] = {lf.z, lf.w, own.x, own.y, own.z, own.w, rt.x};
		float r[2];
#pragma unroll
		for (int j = 0; j < 2; j++) {
			// centre x + 2 j = v[2 + 2 j]; taps centre - R .. centre + R, first tap assigns (ConvolveDownNoBorderUnrolled_F32_F32)
			float total = v[2 + 2 * j - R] * P.k[0];
#pragma unroll
			for (int i = 1; i < KW; i++) total += v[2 + 2 * j - R + i] * P.k[i];
			r[j] = total;
		}
		float* dst = outImg + (long long)y * P.outStride + D0;
		if (w0 && w1) *reinterpret_cast<float2*>(dst) = make_float2(r[0], r[1]);
		else { if (w0) dst[0] = r[0]; if (w1) dst[1] = r[1]; }
	};
#pragma unroll
	for (int q = 0; q < NB; q++) fetch(own[q], edge[q], y0 + q);
	for (int y = y0; y < yEnd; y += NB) {
#pragma unroll
		for (int q = 0; q < NB; q++) {
			emit(own[q], edge[q], y + q);
			fetch(own[q], edge[q], y + q + NB);
		}
	}
}
// Vertical: lane l owns columns 4l..4l+3; a wave produces CD_ROWS_V output rows, walking the 2 CD_ROWS_V + KW - 2 input rows it needs once
// through a register window of KW rows (two new rows per output).
#define CD_ROWS_V 8
template <int KW>
__global__ __launch_bounds__(256) void k_conv_down_v_stream(ConvDownParams P) {
	constexpr int R = KW / 2;
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int x = blockIdx.x * 256 + 4 * lane;
	const int outH = P.height / 2;
	const int o0 = (blockIdx.y * 4 + wave) * CD_ROWS_V;
	if (x >= P.width || o0 >= outH) return;
	const float* img = P.in + (long long)blockIdx.z * P.inImageStride;
	float* outImg = P.out + (long long)blockIdx.z * P.outImageStride + x;
	const int oEnd = min(o0 + CD_ROWS_V, outH);
	const bool full4 = x + 3 < P.width;
	auto loadRow = [&](int yy) -> float4 {
		if (yy < 0 || yy >= P.height) return make_float4(0, 0, 0, 0);   // only ever feeds outputs that are not interior
		return loadRow4(img + (long long)yy * P.inStride, x, P.width);
	};
	// win[i] = input row 2 o - R + i of the current output o
	float4 win[KW];
#pragma unroll
	for (int i = 0; i < KW; i++) win[i] = loadRow(2 * o0 - R + i);
	for (int o = o0; o < oEnd; o++) {
		float4 n0 = make_float4(0, 0, 0, 0), n1 = n0;
		if (o + 1 < oEnd) { n0 = loadRow(2 * o + R + 1); n1 = loadRow(2 * o + R + 2); }   // the two rows the next output adds
		if (downInterior(P, o)) {
			float r[4] = {win[0].x * P.k[0], win[0].y * P.k[0], win[0].z * P.k[0], win[0].w * P.k[0]};
#pragma unroll
			for (int i = 1; i < KW; i++) { r[0] += win[i].x * P.k[i]; r[1] += win[i].y * P.k[i]; r[2] += win[i].z * P.k[i]; r[3] += win[i].w * P.k[i]; }
			float* dst = outImg + (long long)o * P.outStride;
			if (full4) *reinterpret_cast<float4*>(dst) = make_float4(r[0], r[1], r[2], r[3]);
			else
				for (int q = 0; q < 4 && x + q < P.width; q++) dst[q] = r[q];
		}
#pragma unroll
		for (int i = 0; i + 2 < KW; i++) win[i] = win[i + 2];
		win[KW - 2] = n0;
		win[KW - 1] = n1;
	}
}

static int downMaxSide(int sideLength, int skip, int radius) {
	int ret = sideLength - (sideLength % skip);
	if (ret + radius >= sideLength) {
		ret = sideLength - radius - 1;
		ret = ret - (ret % skip);
	} else {
		ret -= skip;
	}
	return ret;
}
static int downOffset(int skip, int radius) { return radius <= skip ? skip : radius + radius % skip; }

int bhip_launch_conv_down(bhip_ctx* ctx, bool vertical, const float* kernel, int kw, const float* in, long long inImageStride, int inStride, int width,
						  int height, float* out, long long outImageStride, int outStride, int outWidth, int outHeight, int skip, int batch) {
	if (kw <= 0 || kw > BHIP_MAX_TAPS) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "kernel width not supported");
	// ConvolveImageDownNoBorder.checkParameters* (I:alg/filter/convolve/ConvolveImageDownNoBorder.java:160-186)
	if (skip <= 0) return bhip_fail(ctx, BHIP_ERR_INVALID, "Skip must be >= 1");
	if (outWidth < width / skip) return bhip_fail(ctx, BHIP_ERR_INVALID, "Output width is too small");
	if (outHeight < height / skip) return bhip_fail(ctx, BHIP_ERR_INVALID, "Output height is too small");
	// checkParametersH/V on the no-border path; on the naive path GrayF32.set would throw ImageAccessException for the same shapes
	if (vertical && outWidth < width) return bhip_fail(ctx, BHIP_ERR_INVALID, "Output width is too small");
	if (!vertical && outHeight < height) return bhip_fail(ctx, BHIP_ERR_INVALID, "Output height is too small");
	ConvDownParams P;
	P.in = in; P.out = out; P.inImageStride = inImageStride; P.outImageStride = outImageStride; P.inStride = inStride; P.outStride = outStride;
	P.width = width; P.height = height; P.skip = skip; P.kw = kw; P.radius = kw / 2;
	for (int i = 0; i < kw; i++) P.k[i] = kernel[i];
	const int r = P.radius;
	const int side = vertical ? height : width;
	P.naive = kw >= width ? 1 : 0;   // sic: the vertical form tests the image WIDTH as well
	P.unrolled = (kw % 2 == 1 && (kw == 3 || kw == 5 || kw == 7 || kw == 9 || kw == 11)) ? 1 : 0;
	P.offset = downOffset(skip, r);
	P.offsetRem = P.offset % skip;
	P.maxSide = downMaxSide(side, skip, r);
	P.offsetEnd = P.maxSide + skip;
	P.sideTrunc = side - side % skip;
	if (!P.naive) {
		if (kw % 2 != 1) return bhip_fail(ctx, BHIP_ERR_INVALID, "Non symmetric odd kernels not supported");
		// every loop of the reference must stay inside the image along the filtered axis (Java would throw
		// ArrayIndexOutOfBounds or silently read the neighbouring row)
		bool ok = true;
		if (P.offset <= P.maxSide) ok = ok && P.offset - r >= 0;
		int lastLeft = ((P.offset - 1) / skip) * skip;                        // largest multiple of skip below offset
		ok = ok && lastLeft + r < side;
		if (P.offsetEnd < P.sideTrunc) ok = ok && P.offsetEnd - r >= 0;
		if (!ok) return bhip_fail(ctx, BHIP_ERR_INVALID, "kernel does not fit the image along the filtered axis");
	}
	const int gw = vertical ? width : width / skip;
	const int gh = vertical ? height / skip : height;
	if (gw <= 0 || gh <= 0 || batch <= 0) return BHIP_OK;
	ProfScope prof(ctx, vertical ? "k_conv_down_v" : "k_conv_down_h", 4.0 * batch * ((double)width * height + (double)gw * gh));
	dim3 grid((gw + 255) / 256, gh, batch);
	P.skipInterior = 0;
	// skip 2, unrolled widths 3 / 5, 16-byte aligned rows: the interior goes through the streaming kernels, the general kernel adds the borders
	const bool stream = !P.naive && skip == 2 && P.unrolled && (kw == 3 || kw == 5) && P.offsetRem == 0 && aligned16(in) && aligned16(out) &&
						inStride % 4 == 0 && inImageStride % 4 == 0 && outStride % 4 == 0 && outImageStride % 4 == 0;
	if (stream) {
		if (vertical) {
			dim3 g((width + 255) / 256, (gh + 4 * CD_ROWS_V - 1) / (4 * CD_ROWS_V), batch);
			if (kw == 3) hipLaunchKernelGGL(k_conv_down_v_stream<3>, g, dim3(256), 0, ctx->stream, P);
			else hipLaunchKernelGGL(k_conv_down_v_stream<5>, g, dim3(256), 0, ctx->stream, P);
		} else {
			dim3 g((width + 255) / 256, (height + 4 * CD_ROWS - 1) / (4 * CD_ROWS), batch);
			if (kw == 3) hipLaunchKernelGGL(k_conv_down_h_stream<3>, g, dim3(256), 0, ctx->stream, P);
			else hipLaunchKernelGGL(k_conv_down_h_stream<5>, g, dim3(256), 0, ctx->stream, P);
		}
		P.skipInterior = 1;
	}
	if (stream) {
		// the few border outputs of the filtered axis (see k_conv_down, skipInterior)
		const int outSide = (vertical ? height : width) / skip;
		const int nl = std::min((P.offset + skip - 1) / skip, outSide), dR0 = std::max(nl, std::min((P.offsetEnd + skip - 1) / skip, outSide));
		const int nb = nl + (outSide - dR0);
		if (nb > 0) {
			if (vertical) hipLaunchKernelGGL(k_conv_down<true>, dim3((gw + 255) / 256, nb, batch), dim3(256), 0, ctx->stream, P);
			else hipLaunchKernelGGL(k_conv_down<false>, dim3((unsigned)(((long long)height * nb + 255) / 256), 1, batch), dim3(256), 0, ctx->stream, P);
		}
	} else if (vertical) hipLaunchKernelGGL(k_conv_down<true>, grid, dim3(256), 0, ctx->stream, P);
	else hipLaunchKernelGGL(k_conv_down<false>, grid, dim3(256), 0, ctx->stream, P);
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}

// ---- one pyramid layer in ONE pass: PyramidDiscreteSampleBlur.process's horizontal.process(prev, temp); vertical.process(temp, layer)
// (I:alg/transform/pyramid/PyramidDiscreteSampleBlur.java:88-118) for the usual step -- skip 2, widths 3 / 5.  Same scheme as
// k_blur_fused: a wave owns 256 input columns (128 outputs) x PL_ROWS output rows, walks the input rows once, forms the horizontally
// down-convolved row in registers (the value the reference leaves in `temp`: lane = input columns 4l..4l+3 = outputs 2l, 2l+1) and
// feeds a register ring from which every second input row yields one output row.  `temp` never exists: 4 P_in read + P_in written
// (+ 2R re-read rows per strip) instead of 4(P + P/2) + 4(P/2 + P/4).  Position classes per axis as in k_conv_down (skip 2: left border
// = output 0, right border = outputs centred in [offsetEnd, sideTrunc), interior between; every output of the layer is written).
#define PL_ROWS 16
struct PyrLayerParams {
	const float* in; float* out;
	long long inImageStride, outImageStride;
	int inStride, outStride, width, height;   // input size; outputs: width/2 x height/2
	int offsetX, maxSideX, offsetEndX, sideTruncX;
	int offsetY, maxSideY, offsetEndY, sideTruncY;
	float k[16];
};
// 0 = interior, 1 = border (normalised over the taps inside the image), -1 = not an output of this layer
__device__ __forceinline__ int pyrClass(int D, int outSide, int offset, int maxSide, int offsetEnd, int sideTrunc) {
	if (D >= outSide) return -1;
	const int centre = 2 * D;
	if (centre >= offsetEnd && centre < sideTrunc) return 1;
	if (centre < offset) return 1;
	return centre <= maxSide ? 0 : -1;
}
template <int KW>
__global__ __launch_bounds__(256) void k_pyr_layer_fused(PyrLayerParams P) {
	constexpr int R = KW / 2;   // <= 2: the kernel reaches at most two columns into the neighbouring chunks
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int x = blockIdx.x * 256 + 4 * lane;          // first input column of this lane
	const int W = P.width, H = P.height, outW = W / 2, outH = H / 2;
	const int o0 = (blockIdx.y * 4 + wave) * PL_ROWS;
	if (x >= W || o0 >= outH) return;
	const float* img = P.in + (long long)blockIdx.z * P.inImageStride;
	float* outImg = P.out + (long long)blockIdx.z * P.outImageStride;
	const int oEnd = min(o0 + PL_ROWS, outH);
	const int D0 = x >> 1;
	const int c0 = pyrClass(D0, outW, P.offsetX, P.maxSideX, P.offsetEndX, P.sideTruncX), c1 = pyrClass(D0 + 1, outW, P.offsetX, P.maxSideX, P.offsetEndX, P.sideTruncX);
	const int yFirst = 2 * o0 - R, yLast = min(2 * (oEnd - 1) + R, H - 1);   // input rows of the strip (clipped below when read)
	auto fetch = [&](float4 (&buf)[3], int yy) {
		if (yy >= 0 && yy <= yLast) {
			const float* row = img + (long long)yy * P.inStride;
#pragma unroll
			for (int c = 0; c < 3; c++) buf[c] = loadRow4(row, x - 4 + 4 * c, W);
		}
	};
	// the two outputs of this lane in one input row: what ConvolveImageDownNormalized.horizontal leaves in `temp`
	auto hdown = [&](const float4 (&buf)[3]) -> float2 {
		const float v[12] = {buf[0].x, buf[0].y, buf[0].z, buf[0].w, buf[1].x, buf[1].y, buf[1].z, buf[1].w, buf[2].x, buf[2].y, buf[2].z, buf[2].w};
		float r[2];
#pragma unroll
		for (int j = 0; j < 2; j++) {
			float t[KW];
#pragma unroll
			for (int i = 0; i < KW; i++) t[i] = v[4 + 2 * j - R + i];   // centre = input column x + 2 j = v[4 + 2 j]
			const int cls = j == 0 ? c0 : c1;
			r[j] = cls == 0 ? blurTapsInterior<KW>(t, P.k) : blurTapsBorder<KW>(t, P.k, x + 2 * j - R, W);
		}
		return make_float2(r[0], r[1]);
	};
	float2 ring[KW];   // ring[i] = `temp` row (yy - 2R + i) once input row yy has been pushed
#pragma unroll
	for (int i = 0; i < KW; i++) ring[i] = make_float2(0, 0);
	float4 cur[3], nxt[3];
#pragma unroll
	for (int c = 0; c < 3; c++) { cur[c] = make_float4(0, 0, 0, 0); nxt[c] = make_float4(0, 0, 0, 0); }
	fetch(cur, yFirst);
	fetch(nxt, yFirst + 1);
#pragma unroll 1
	for (int yy = yFirst; yy <= 2 * (oEnd - 1) + R; yy++) {
		float2 hrow = make_float2(0, 0);
		if (yy >= 0 && yy < H) hrow = hdown(cur);
#pragma unroll
		for (int c = 0; c < 3; c++) cur[c] = nxt[c];
		fetch(nxt, yy + 2);
#pragma unroll
		for (int i = 0; i + 1 < KW; i++) ring[i] = ring[i + 1];
		ring[KW - 1] = hrow;
		const int cy = yy - R;                   // centre row of the window now in the ring
		if (cy >= 2 * o0 && (cy & 1) == 0) {
			const int o = cy >> 1;
			const int cls = pyrClass(o, outH, P.offsetY, P.maxSideY, P.offsetEndY, P.sideTruncY);   // wave-uniform
			if (cls >= 0) {
				float r[2];
#pragma unroll
				for (int q = 0; q < 2; q++) {
					float tv[KW];
#pragma unroll
					for (int i = 0; i < KW; i++) tv[i] = q == 0 ? ring[i].x : ring[i].y;
					r[q] = cls == 0 ? blurTapsInterior<KW>(tv, P.k) : blurTapsBorder<KW>(tv, P.k, cy - R, H);
				}
				float* dst = outImg + (long long)o * P.outStride + D0;
				if (c0 >= 0 && c1 >= 0) *reinterpret_cast<float2*>(dst) = make_float2(r[0], r[1]);
				else { if (c0 >= 0) dst[0] = r[0]; if (c1 >= 0) dst[1] = r[1]; }
			}
		}
	}
}

// ---------------- batched image copy (pyramid layer 0 at scale 1: ImagePyramidBase keeps a copy of the input) ----------------
__global__ __launch_bounds__(256) void k_copy_images(const float* __restrict__ in, long long inImageStride, int inStride, float* __restrict__ out,
													   long long outImageStride, int outStride, int width, int height, int vec) {
	const int x = (blockIdx.x * 256 + threadIdx.x) * 4;
	if (x >= width) return;
	const float* src = in + (long long)blockIdx.z * inImageStride + x;
	float* dst = out + (long long)blockIdx.z * outImageStride + x;
	for (int y = blockIdx.y; y < height; y += gridDim.y) {
		const float* s = src + (long long)y * inStride;
		float* d = dst + (long long)y * outStride;
		if (vec && x + 3 < width) *reinterpret_cast<float4*>(d) = *reinterpret_cast<const float4*>(s);
		else
			for (int q = 0; q < 4 && x + q < width; q++) d[q] = s[q];
	}
}
// *done = false: the shape / kernel is outside what the fused layer kernel covers (the caller runs the two passes through `temp`)
int bhip_launch_pyr_layer_fused(bhip_ctx* ctx, const float* kernel, int kw, const float* in, long long inImageStride, int inStride, int width, int height,
								float* out, long long outImageStride, int outStride, int skip, int batch, bool* done) {
	*done = false;
	if (skip != 2 || !(kw == 3 || kw == 5) || batch <= 0) return BHIP_OK;
	const int r = kw / 2;
	if (kw >= width / 2 || kw >= height) return BHIP_OK;        // naive forms (the reference's vertical pass tests the width of `temp`, width / 2)
	if (width / 2 <= 0 || height / 2 <= 0) return BHIP_OK;
	if (!(aligned16(in) && inStride % 4 == 0 && inImageStride % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0 && outStride % 2 == 0 && outImageStride % 2 == 0))
		return BHIP_OK;
	if (bhip_env_flag("BHIP_PYRAMID_TWO_PASS")) return BHIP_OK;  // parity cross-check of the two forms
	PyrLayerParams P;
	P.in = in; P.out = out; P.inImageStride = inImageStride; P.outImageStride = outImageStride; P.inStride = inStride; P.outStride = outStride;
	P.width = width; P.height = height;
	for (int i = 0; i < kw; i++) P.k[i] = kernel[i];
	for (int axis = 0; axis < 2; axis++) {
		const int side = axis == 0 ? width : height;
		const int offset = downOffset(2, r), maxSide = downMaxSide(side, 2, r), offsetEnd = maxSide + 2, sideTrunc = side - side % 2;
		if (offset % 2 != 0) return BHIP_OK;
		// the same fit checks bhip_launch_conv_down makes (a shape it rejects must be rejected there, with its message)
		bool ok = true;
		if (offset <= maxSide) ok = ok && offset - r >= 0;
		const int lastLeft = ((offset - 1) / 2) * 2;
		ok = ok && lastLeft + r < side;
		if (offsetEnd < sideTrunc) ok = ok && offsetEnd - r >= 0;
		if (!ok) return BHIP_OK;
		if (axis == 0) { P.offsetX = offset; P.maxSideX = maxSide; P.offsetEndX = offsetEnd; P.sideTruncX = sideTrunc; }
		else { P.offsetY = offset; P.maxSideY = maxSide; P.offsetEndY = offsetEnd; P.sideTruncY = sideTrunc; }
	}
	// algorithmic bytes: the two passes of SURVEY 8d, 4 (P_in + P_out) each
	const double Pin = (double)width * height, Ptmp = (double)(width / 2) * height, Pout = (double)(width / 2) * (height / 2);
	ProfScope prof(ctx, "k_pyr_layer_fused", 4.0 * batch * ((Pin + Ptmp) + (Ptmp + Pout)));
	dim3 grid((width + 255) / 256, (height / 2 + 4 * PL_ROWS - 1) / (4 * PL_ROWS), batch);
	if (kw == 3) hipLaunchKernelGGL(k_pyr_layer_fused<3>, grid, dim3(256), 0, ctx->stream, P);
	else hipLaunchKernelGGL(k_pyr_layer_fused<5>, grid, dim3(256), 0, ctx->stream, P);
	BHIP_HIP(ctx, hipGetLastError());
	*done = true;
	return BHIP_OK;
}

int bhip_launch_copy_images(bhip_ctx* ctx, const float* in, long long inImageStride, int inStride, float* out, long long outImageStride, int outStride,
							int width, int height, int batch) {
	if (width <= 0 || height <= 0 || batch <= 0) return BHIP_OK;
	const int vec = aligned16(in) && aligned16(out) && inStride % 4 == 0 && outStride % 4 == 0 && inImageStride % 4 == 0 && outImageStride % 4 == 0;
	dim3 grid((width + 1023) / 1024, std::min(height, 256), batch);
	hipLaunchKernelGGL(k_copy_images, grid, dim3(256), 0, ctx->stream, in, inImageStride, inStride, out, outImageStride, outStride, width, height, vec);
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}

// ---------------- gradients ----------------
struct GradParams {
	const float* in;
	float* dx;
	float* dy;
	long long inImageStride, outImageStride;   // floats between the images of a batch
	int inStride, outStride, width, height;
	int border;  // 0: frame untouched, 1: ImageBorderValue(0)
};

__device__ __forceinline__ float at0(const GradParams& P, const float* img, int x, int y) {
	return (x >= 0 && x < P.width && y >= 0 && y < P.height) ? img[(long long)y * P.inStride + x] : 0.0f;
}

// One pixel of GradientSobel / GradientThree from its 3x3 neighbourhood a[row][col] (0 outside the image), reference expressions:
//   Sobel interior   GradientSobel_UnrolledOuter.process_F32_sub; frame with a border = generic border convolution with kernelDerivX/Y_F32
//                    (total = 0; total += get(x+j,y+i)*k in row-major kernel order)
//   three-tap        GradientThree_Standard.process inside; with a border the first/last column (row) of derivX (derivY) takes the generic
//                    form, rows {0,1,H-2,H-1} (columns for derivY) the unrolled 3-tap form
// Returns false when the pixel is left untouched (frame pixel, no border policy).
template <int KIND>
__device__ __forceinline__ bool gradOne(const float a[3][3], int x, int y, int W, int H, int border, float& dx, float& dy) {
	const bool interior = x >= 1 && x < W - 1 && y >= 1 && y < H - 1;
	if (KIND == 0) {
		if (interior) {
			const float v = (a[2][2] - a[0][0]) * 0.25f;
			const float w = (a[2][0] - a[0][2]) * 0.25f;
			dy = (a[2][1] - a[0][1]) * 0.5f + v + w;
			dx = (a[1][2] - a[1][0]) * 0.5f + v - w;
			return true;
		}
		if (!border) return false;
		const float kx[9] = {-0.25f, 0, 0.25f, -0.5f, 0, 0.5f, -0.25f, 0, 0.25f};
		const float ky[9] = {-0.25f, -0.5f, -0.25f, 0, 0, 0, 0.25f, 0.5f, 0.25f};
		float tx = 0, ty = 0;
#pragma unroll
		for (int i = 0; i < 3; i++)
#pragma unroll
			for (int j = 0; j < 3; j++) {
				tx += a[i][j] * kx[i * 3 + j];
				ty += a[i][j] * ky[i * 3 + j];
			}
		dx = tx; dy = ty;
		return true;
	}
	if (!border) {
		if (!interior) return false;
		dx = (a[1][2] - a[1][0]) * 0.5f;
		dy = (a[2][1] - a[0][1]) * 0.5f;
		return true;
	}
	const float k0 = -0.5f, k1 = 0.0f, k2 = 0.5f;
	if (x == 0 || x == W - 1) {
		float t = 0;
		t += a[1][0] * k0; t += a[1][1] * k1; t += a[1][2] * k2;
		dx = t;
	} else if (y <= 1 || y >= H - 2) {
		float t = a[1][0] * k0; t += a[1][1] * k1; t += a[1][2] * k2;
		dx = t;
	} else {
		dx = (a[1][2] - a[1][0]) * 0.5f;
	}
	if (y == 0 || y == H - 1) {
		float t = 0;
		t += a[0][1] * k0; t += a[1][1] * k1; t += a[2][1] * k2;
		dy = t;
	} else if (x <= 1 || x >= W - 2) {
		float t = a[0][1] * k0; t += a[1][1] * k1; t += a[2][1] * k2;
		dy = t;
	} else {
		dy = (a[2][1] - a[0][1]) * 0.5f;
	}
	return true;
}

// General form: one pixel per thread, any alignment
template <int KIND>
__global__ __launch_bounds__(256) void k_grad(GradParams P) {
	const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
	if (x >= P.width) return;
	const float* img = P.in + (long long)blockIdx.z * P.inImageStride;
	float a[3][3];
#pragma unroll
	for (int i = 0; i < 3; i++)
#pragma unroll
		for (int j = 0; j < 3; j++) a[i][j] = at0(P, img, x + j - 1, y + i - 1);
	float dx, dy;
	if (!gradOne<KIND>(a, x, y, P.width, P.height, P.border, dx, dy)) return;
	const long long o = (long long)blockIdx.z * P.outImageStride + (long long)y * P.outStride + x;
	P.dx[o] = dx;
	P.dy[o] = dy;
}

// Streaming form: a wave walks GR_ROWS rows of a 256-column strip; lane l owns columns 4l .. 4l+3 and keeps a three-row window in
// registers, so every input row is loaded once per strip (16 bytes per lane); the two columns beside a lane's four come from the
// neighbouring lanes, the strip's outermost two from single loads.  Outputs leave as 16-byte stores.  HBM: 4P read + 8P written.
#define GR_ROWS 8
struct GradRow { float v[6]; };   // columns x-1 .. x+4 of one row (0 outside the image)
__device__ __forceinline__ GradRow gradLoadRow(const GradParams& P, const float* img, int x, int y, int lane) {
	GradRow r;
	if (y < 0 || y >= P.height) {
#pragma unroll
		for (int i = 0; i < 6; i++) r.v[i] = 0.0f;
		return r;
	}
	const float* row = img + (long long)y * P.inStride;
	const float4 c = x < P.width ? loadRow4(row, x, P.width) : make_float4(0, 0, 0, 0);
	float left = __shfl_up(c.w, 1, 64), right = __shfl_down(c.x, 1, 64);
	if (lane == 0) left = (x - 1 >= 0 && x - 1 < P.width) ? row[x - 1] : 0.0f;
	if (lane == 63) right = (x + 4 < P.width) ? row[x + 4] : 0.0f;
	r.v[0] = left; r.v[1] = c.x; r.v[2] = c.y; r.v[3] = c.z; r.v[4] = c.w; r.v[5] = right;
	return r;
}
template <int KIND>
__global__ __launch_bounds__(256) void k_grad_stream(GradParams P) {
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int x = blockIdx.x * 256 + 4 * lane;
	const int y0 = (blockIdx.y * 4 + wave) * GR_ROWS;
	if (y0 >= P.height) return;
	const float* img = P.in + (long long)blockIdx.z * P.inImageStride;
	float* dxImg = P.dx + (long long)blockIdx.z * P.outImageStride;
	float* dyImg = P.dy + (long long)blockIdx.z * P.outImageStride;
	GradRow r0 = gradLoadRow(P, img, x, y0 - 1, lane), r1 = gradLoadRow(P, img, x, y0, lane), r2 = gradLoadRow(P, img, x, y0 + 1, lane);
	const int yEnd = min(y0 + GR_ROWS, P.height);
	for (int y = y0; y < yEnd; y++) {
		// row y+2 (the bottom row of the NEXT output) is requested before this output is computed: one row of loads always in flight
		GradRow r3 = r2;
		if (y + 1 < yEnd) r3 = gradLoadRow(P, img, x, y + 2, lane);
		if (x < P.width) {
			float dx[4], dy[4];
			bool wr[4];
#pragma unroll
			for (int j = 0; j < 4; j++) {
				const float a[3][3] = {{r0.v[j], r0.v[j + 1], r0.v[j + 2]}, {r1.v[j], r1.v[j + 1], r1.v[j + 2]}, {r2.v[j], r2.v[j + 1], r2.v[j + 2]}};
				wr[j] = gradOne<KIND>(a, x + j, y, P.width, P.height, P.border, dx[j], dy[j]);
			}
			const long long o = (long long)y * P.outStride + x;
			if (x + 3 < P.width && wr[0] && wr[1] && wr[2] && wr[3]) {
				*reinterpret_cast<float4*>(dxImg + o) = make_float4(dx[0], dx[1], dx[2], dx[3]);
				*reinterpret_cast<float4*>(dyImg + o) = make_float4(dy[0], dy[1], dy[2], dy[3]);
			} else {
#pragma unroll
				for (int j = 0; j < 4; j++)
					if (x + j < P.width && wr[j]) { dxImg[o + j] = dx[j]; dyImg[o + j] = dy[j]; }
			}
		}
		r0 = r1; r1 = r2; r2 = r3;
	}
}

int bhip_launch_gradient(bhip_ctx* ctx, int kind, const float* in, int inStride, int width, int height, float* dx, float* dy, int outStride, int border,
						 int batch, long long inImageStride, long long outImageStride) {
	if (width <= 0 || height <= 0 || batch <= 0) return BHIP_OK;
	GradParams P{in, dx, dy, inImageStride, outImageStride, inStride, outStride, width, height, border};
	ProfScope prof(ctx, kind == 0 ? "k_sobel" : "k_three", 12.0 * width * height * batch);
	const bool stream = aligned16(in) && aligned16(dx) && aligned16(dy) && inStride % 4 == 0 && outStride % 4 == 0 && inImageStride % 4 == 0 &&
						outImageStride % 4 == 0;
	if (stream) {
		dim3 grid((width + 255) / 256, (height + 4 * GR_ROWS - 1) / (4 * GR_ROWS), batch);
		if (kind == 0) hipLaunchKernelGGL(k_grad_stream<0>, grid, dim3(256), 0, ctx->stream, P);
		else hipLaunchKernelGGL(k_grad_stream<1>, grid, dim3(256), 0, ctx->stream, P);
	} else {
		dim3 grid((width + 255) / 256, height, batch);
		if (kind == 0) hipLaunchKernelGGL(k_grad<0>, grid, dim3(256), 0, ctx->stream, P);
		else hipLaunchKernelGGL(k_grad<1>, grid, dim3(256), 0, ctx->stream, P);
	}
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}

// GradientToEdgeFeatures.intensityE / intensityAbs (F:alg/feature/detect/edge/impl/ImplGradientToEdgeFeatures.java:40-85) and the squared
// magnitude dx*dx + dy*dy (no reference function of its own: the |grad|^2 of BASELINE config 5; same products and sum as intensityE,
// without the square root).  Element-wise, 8P read + 4P written.
__global__ __launch_bounds__(256) void k_grad_intensity(const float* __restrict__ dx, const float* __restrict__ dy, long long dImageStride, int dStride,
														 float* __restrict__ out, long long oImageStride, int oStride, int width, int height, int kind) {
	const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
	if (x >= width) return;
	const long long i = (long long)blockIdx.z * dImageStride + (long long)y * dStride + x;
	const float a = dx[i], b = dy[i];
	float r;
	if (kind == 0) r = sqrtf(a * a + b * b);           // (float)Math.sqrt((double)f): correctly rounded, as sqrtf
	else if (kind == 1) r = fabsf(a) + fabsf(b);
	else r = a * a + b * b;
	out[(long long)blockIdx.z * oImageStride + (long long)y * oStride + x] = r;
}
int bhip_launch_grad_intensity(bhip_ctx* ctx, int kind, const float* dx, const float* dy, long long dImageStride, int dStride, float* out,
							   long long oImageStride, int oStride, int width, int height, int batch) {
	if (kind < 0 || kind > 2) return bhip_fail(ctx, BHIP_ERR_INVALID, "unknown gradient intensity");
	if (width <= 0 || height <= 0 || batch <= 0) return BHIP_OK;
	ProfScope prof(ctx, "k_grad_intensity", 12.0 * width * height * batch);
	hipLaunchKernelGGL(k_grad_intensity, dim3((width + 255) / 256, height, batch), dim3(256), 0, ctx->stream, dx, dy, dImageStride, dStride, out, oImageStride,
					   oStride, width, height, kind);
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}

// ---------------- BRIEF ----------------
struct BriefParams {
	const float* img;
	long long imageStride;    // elements between the images of a batch
	int stride, width, height, radius, numPoints, words, n;
	const int* samplePoints;  // [numPoints][2]
	const int* compare;       // [numPoints][2]
	const double* xy;         // [n][xyStride] compact, or [image][xyImageStride] records when xyImageStride != 0; (x, y) first
	int* out;                 // [n][words]
	const int* start;         // batched: points of image b are [start[b], start[b+1]); nullptr: one image, n points
	int xyStride;             // doubles between consecutive points (2 for (x,y) pairs, 4 for KeyPoint records)
	long long xyImageStride;  // doubles between the point lists of consecutive images (0: one compact list indexed by start[])
};

// one thread = one 32-pair word of one key point.  PIX = float (ImplDescribeBinaryCompare_F32: a pair outside the image is skipped
// without shifting the word) or unsigned char (ImplDescribeBinaryCompare_U8.java:73-101: the word is shifted for EVERY pair)
template <class PIX>
__global__ __launch_bounds__(256) void k_brief(BriefParams P) {
	const int first = P.start ? P.start[blockIdx.y] : 0;
	const int count = P.start ? P.start[blockIdx.y + 1] - first : P.n;
	const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (t >= (long long)count * P.words) return;
	const int pl = (int)(t / P.words), word = (int)(t - (long long)pl * P.words);
	const int p = first + pl;
	const double* pt = P.xyImageStride ? P.xy + (long long)blockIdx.y * P.xyImageStride + (long long)pl * P.xyStride : P.xy + (long long)p * P.xyStride;
	const int c_x = (int)pt[0], c_y = (int)pt[1];
	const bool inside = !(c_x - P.radius < 0 || c_x + P.radius >= P.width || c_y - P.radius < 0 || c_y + P.radius >= P.height);
	const int i0 = word * 32, i1 = min(P.numPoints, i0 + 32);
	const PIX* img = (const PIX*)P.img + (long long)blockIdx.y * P.imageStride;
	unsigned int desc = 0;
	for (int j = i0; j < i1; j++) {
		const int ia = P.compare[2 * j], ib = P.compare[2 * j + 1];
		const int ax = P.samplePoints[2 * ia] + c_x, ay = P.samplePoints[2 * ia + 1] + c_y;
		const int bx = P.samplePoints[2 * ib] + c_x, by = P.samplePoints[2 * ib + 1] + c_y;
		const bool ok = inside || (ax >= 0 && ax < P.width && ay >= 0 && ay < P.height && bx >= 0 && bx < P.width && by >= 0 && by < P.height);
		if (sizeof(PIX) == 1) {
			desc = desc * 2u;
			if (ok && img[(long long)ay * P.stride + ax] < img[(long long)by * P.stride + bx]) desc += 1u;
		} else if (ok) {
			const PIX va = img[(long long)ay * P.stride + ax];
			const PIX vb = img[(long long)by * P.stride + bx];
			desc = desc * 2u + (va < vb ? 1u : 0u);
		}
	}
	P.out[(long long)p * P.words + word] = (int)desc;
}

// LDS-patch form (the normal path): one wavefront per key point.  The (2R+1)^2 neighbourhood of the point is staged once with row-wise
// loads (33 rows of 132 bytes for R = 16) and the 2 x numPoints samples are LDS reads -- against two scattered global reads per pair in
// the per-word kernel above.  The pair table is resolved once per workgroup into patch offsets (and the raw sample coordinates, for
// the border rules).  Interior points and every GrayU8 point: lane l takes pairs 8l .. 8l+7 of each run of 512, builds its byte of the
// word, the four lanes of a quad OR their bytes together (bit of pair j in a word of `cnt` pairs: cnt - 1 - j, i.e. the first pair ends
// up in the top bit of a full word -- the reference's shift-and-add loop).  GrayF32 border points skip pairs without shifting, so the bit
// position depends on the pairs before: one lane per word walks its pairs in order, still from the patch.
#define BRIEF_KP_PER_WAVE 4
template <class PIX>
__global__ __launch_bounds__(256) void k_brief_patch(BriefParams P) {
	extern __shared__ __attribute__((aligned(16))) unsigned char briefLds[];
	const int R = P.radius, PW = 2 * R + 1, PP = PW * PW;
	unsigned int* offs = (unsigned int*)briefLds;                 // [numPoints] patch offset of sample A | patch offset of sample B << 16
	unsigned int* xy8 = offs + P.numPoints;                       // [numPoints] (ax, ay, bx, by) as signed bytes
	float* patches = (float*)(xy8 + P.numPoints);                 // [4 waves][PP]
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	for (int j = threadIdx.x; j < P.numPoints; j += blockDim.x) {
		const int ia = P.compare[2 * j], ib = P.compare[2 * j + 1];
		const int ax = P.samplePoints[2 * ia], ay = P.samplePoints[2 * ia + 1], bx = P.samplePoints[2 * ib], by = P.samplePoints[2 * ib + 1];
		offs[j] = (unsigned)((ay + R) * PW + ax + R) | ((unsigned)((by + R) * PW + bx + R) << 16);
		xy8[j] = (unsigned)(ax & 0xff) | ((unsigned)(ay & 0xff) << 8) | ((unsigned)(bx & 0xff) << 16) | ((unsigned)(by & 0xff) << 24);
	}
	__syncthreads();
	const int first = P.start ? P.start[blockIdx.y] : 0;
	const int count = P.start ? P.start[blockIdx.y + 1] - first : P.n;
	const PIX* __restrict__ img = (const PIX*)P.img + (long long)blockIdx.y * P.imageStride;
	float* patch = patches + wave * PP;
	const int W = P.width, H = P.height;
	for (int it = 0; it < BRIEF_KP_PER_WAVE; it++) {
		const int pl = (blockIdx.x * 4 + wave) * BRIEF_KP_PER_WAVE + it;   // wave-uniform
		if (pl >= count) break;
		const int p = first + pl;
		const double* pt = P.xyImageStride ? P.xy + (long long)blockIdx.y * P.xyImageStride + (long long)pl * P.xyStride : P.xy + (long long)p * P.xyStride;
		const int c_x = (int)pt[0], c_y = (int)pt[1];
		const bool inside = !(c_x - R < 0 || c_x + R >= W || c_y - R < 0 || c_y + R >= H);
		__builtin_amdgcn_wave_barrier();
		for (int i = lane; i < PP; i += 64) {
			const int py = i / PW, px = i - py * PW;
			const int gx = c_x - R + px, gy = c_y - R + py;
			float v = 0.f;
			if (inside || (gx >= 0 && gx < W && gy >= 0 && gy < H)) v = (float)img[(long long)gy * P.stride + gx];
			patch[i] = v;
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		int* out = P.out + (long long)p * P.words;
		if (inside || sizeof(PIX) == 1) {
			for (int base = 0; base < P.numPoints; base += 512) {
				const int j0 = base + lane * 8;
				const int wordBase = j0 & ~31;
				const int cnt = min(32, P.numPoints - wordBase);   // pairs in this lane's word
				unsigned int bits = 0;
#pragma unroll
				for (int q = 0; q < 8; q++) {
					const int j = j0 + q;
					if (j < P.numPoints) {
						const unsigned int o = offs[j];
						bool lt = patch[o & 0xffffu] < patch[o >> 16];
						if (!inside) {
							const unsigned int c = xy8[j];
							const int ax = (int)(signed char)(c & 0xff) + c_x, ay = (int)(signed char)((c >> 8) & 0xff) + c_y;
							const int bx = (int)(signed char)((c >> 16) & 0xff) + c_x, by = (int)(signed char)(c >> 24) + c_y;
							lt = lt && ax >= 0 && ax < W && ay >= 0 && ay < H && bx >= 0 && bx < W && by >= 0 && by < H;
						}
						bits |= (lt ? 1u : 0u) << (cnt - 1 - (j - wordBase));
					}
				}
				bits |= __shfl_xor(bits, 1, 64);
				bits |= __shfl_xor(bits, 2, 64);
				if ((lane & 3) == 0 && wordBase < P.numPoints) out[wordBase >> 5] = (int)bits;
			}
		} else {
			// ImplDescribeBinaryCompare_F32.processBorder: a pair with a sample outside the frame is skipped WITHOUT shifting the word
			for (int word = lane; word < P.words; word += 64) {
				const int i0 = word * 32, i1 = min(P.numPoints, i0 + 32);
				unsigned int desc = 0;
				for (int j = i0; j < i1; j++) {
					const unsigned int c = xy8[j];
					const int ax = (int)(signed char)(c & 0xff) + c_x, ay = (int)(signed char)((c >> 8) & 0xff) + c_y;
					const int bx = (int)(signed char)((c >> 16) & 0xff) + c_x, by = (int)(signed char)(c >> 24) + c_y;
					if (ax >= 0 && ax < W && ay >= 0 && ay < H && bx >= 0 && bx < W && by >= 0 && by < H) {
						const unsigned int o = offs[j];
						desc = desc * 2u + (patch[o & 0xffffu] < patch[o >> 16] ? 1u : 0u);
					}
				}
				out[word] = (int)desc;
			}
		}
	}
}

// start == nullptr: n points on one image.  Otherwise `batch` images and device prefix `start` (batch+1); maxCount = largest per-image count.
int bhip_launch_brief(bhip_ctx* ctx, const float* img, int stride, int width, int height, int radius, int numPoints, const int* samplePoints,
					  const int* compare, const double* xy, int n, int* out, bool u8, int batch, long long imageStride, const int* start, int maxCount,
					  int xyStride, long long xyImageStride, bool patchOk) {
	if (n <= 0) return BHIP_OK;
	BriefParams P{img, imageStride, stride, width, height, radius, numPoints, (numPoints + 31) / 32, n, samplePoints, compare, xy, out, start, xyStride, xyImageStride};
	const long long total = (long long)(start ? maxCount : n) * P.words;
	if (total <= 0) return BHIP_OK;
	dim3 grid((unsigned)((total + 255) / 256), start ? batch : 1);
	// per point: the (2R+1)^2 patch read once + the words written
	ProfScope prof(ctx, "k_brief", (double)n * ((double)(2 * radius + 1) * (2 * radius + 1) * (u8 ? 1 : 4) + 4.0 * P.words));
	const size_t patchLds = (size_t)numPoints * 8 + (size_t)4 * (2 * radius + 1) * (2 * radius + 1) * 4;
	if (patchOk && radius <= 40 && numPoints <= 4096 && patchLds <= 64 * 1024 && !bhip_env_flag("BHIP_BRIEF_GATHER")) {
		const long long pts = start ? maxCount : n;
		dim3 pgrid((unsigned)((pts + 4 * BRIEF_KP_PER_WAVE - 1) / (4 * BRIEF_KP_PER_WAVE)), start ? batch : 1);
		if (u8) hipLaunchKernelGGL(k_brief_patch<unsigned char>, pgrid, dim3(256), patchLds, ctx->stream, P);
		else hipLaunchKernelGGL(k_brief_patch<float>, pgrid, dim3(256), patchLds, ctx->stream, P);
	} else if (u8) hipLaunchKernelGGL(k_brief<unsigned char>, grid, dim3(256), 0, ctx->stream, P);
	else hipLaunchKernelGGL(k_brief<float>, grid, dim3(256), 0, ctx->stream, P);
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}


// ---------------- band average of a planar image ----------------
// ImplConvertPlanarToGray.average(Planar<GrayF32>, GrayF32)  I:core/image/impl/ImplConvertPlanarToGray.java:296-336:
// one band: copy; three bands: sum = b0; sum += b1; sum += b2; sum / 3; otherwise sum = 0; sum += b_i ...; sum / numBands
__global__ __launch_bounds__(256) void k_planar_average(const float* __restrict__ bands, long long bandStride, int numBands, long long n, float* __restrict__ out) {
	const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	float r;
	if (numBands == 1) {
		r = bands[i];
	} else if (numBands == 3) {
		float sum = bands[i];
		sum += bands[bandStride + i];
		sum += bands[2 * bandStride + i];
		r = sum / 3;
	} else {
		float sum = 0;
		for (int b = 0; b < numBands; b++) sum += bands[b * bandStride + i];
		r = sum / numBands;
	}
	out[i] = r;
}
int bhip_launch_planar_average(bhip_ctx* ctx, const float* bands, long long bandStride, int numBands, long long n, float* out) {
	if (n <= 0) return BHIP_OK;
	ProfScope prof(ctx, "k_planar_average", 4.0 * n * (numBands + 1));
	hipLaunchKernelGGL(k_planar_average, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, bands, bandStride, numBands, n, out);
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}


// ---------------- gradient corner intensity (Shi-Tomasi / Harris on box-window sums of the gradient products) ----------------
// GradientCornerIntensity.process = ImplSsdCornerBox.process   F:alg/feature/detect/intensity/impl/ImplSsdCornerBox.java:36-51
//   horizontal() / vertical()                                  F:alg/feature/detect/intensity/impl/ImplSsdCorner_F32.java:62-127, 133-196
//   scores                                                     F:.../impl/ShiTomasiCorner_F32.java:33-42, HarrisCorner_F32.java:45-50
// The reference's window sums are running float sums (subtract the sample leaving the window, add the one entering), so every value
// depends on the history of its whole row (horizontal pass) or column (vertical pass).  The chains are kept exactly: one thread walks
// one row, then one thread walks one column; parallelism is across rows / columns (and images).  Bound: the horizontal pass is a
// latency chain over W per row (each lane streams its own row through L1); the vertical pass is coalesced and HBM bound (12P read, 4P write).
struct CornerParams {
	const float* dx; const float* dy;
	long long dImageStride, hImageStride, iImageStride;   // floats between the images of a batch (blockIdx.y)
	int dStride, width, height, radius;
	float* hXX; float* hXY; float* hYY;   // dense width x height planes
	float* intensity; int iStride;
	int kind; float kappa;
};

__global__ __launch_bounds__(64) void k_corner_rows(CornerParams P) {
	const int row = blockIdx.x * blockDim.x + threadIdx.x;
	if (row >= P.height) return;
	const int W = P.width, r = P.radius, ww = 2 * r + 1;
	const float* __restrict__ X = P.dx + (long long)blockIdx.y * P.dImageStride + (long long)row * P.dStride;
	const float* __restrict__ Y = P.dy + (long long)blockIdx.y * P.dImageStride + (long long)row * P.dStride;
	const long long ho = (long long)blockIdx.y * P.hImageStride + (long long)row * W;
	float* oXX = P.hXX + ho; float* oXY = P.hXY + ho; float* oYY = P.hYY + ho;
	float tXX = 0, tXY = 0, tYY = 0;
	for (int i = 0; i < ww; i++) {
		const float dx = X[i], dy = Y[i];
		tXX += dx * dx; tXY += dx * dy; tYY += dy * dy;
	}
	oXX[r] = tXX; oXY[r] = tXY; oYY[r] = tYY;
	int i = ww;
	// eight steps at a time: the 32 loads of a batch are independent and issued together, the chain arithmetic keeps its order
	for (; i + 8 <= W; i += 8) {
		float ax[8], ay[8], bx[8], by[8], rXX[8], rXY[8], rYY[8];
#pragma unroll
		for (int k = 0; k < 8; k++) { ax[k] = X[i - ww + k]; ay[k] = Y[i - ww + k]; bx[k] = X[i + k]; by[k] = Y[i + k]; }
#pragma unroll
		for (int k = 0; k < 8; k++) {
			tXX -= ax[k] * ax[k]; tXY -= ax[k] * ay[k]; tYY -= ay[k] * ay[k];
			tXX += bx[k] * bx[k]; tXY += bx[k] * by[k]; tYY += by[k] * by[k];
			rXX[k] = tXX; rXY[k] = tXY; rYY[k] = tYY;
		}
#pragma unroll
		for (int k = 0; k < 8; k++) { oXX[i - r + k] = rXX[k]; oXY[i - r + k] = rXY[k]; oYY[i - r + k] = rYY[k]; }
	}
	for (; i < W; i++) {
		float dx = X[i - ww], dy = Y[i - ww];
		tXX -= dx * dx; tXY -= dx * dy; tYY -= dy * dy;
		dx = X[i]; dy = Y[i];
		tXX += dx * dx; tXY += dx * dy; tYY += dy * dy;
		oXX[i - r] = tXX; oXY[i - r] = tXY; oYY[i - r] = tYY;
	}
}

__device__ __forceinline__ float cornerScore(int kind, float kappa, float xx, float xy, float yy) {
	if (kind == 0) {
		const float left = (xx + yy) * 0.5f;
		const float b = (xx - yy) * 0.5f;
		const float right = sqrtf(b * b + xy * xy);   // (float)Math.sqrt((double)f) == correctly rounded float sqrt
		return left - right;
	}
	const float trace = xx + yy;
	return (xx * yy - xy * xy) - kappa * trace * trace;
}

__global__ __launch_bounds__(256) void k_corner_cols(CornerParams P) {
	const int W = P.width, H = P.height, r = P.radius, kw = 2 * r + 1;
	const int x = r + blockIdx.x * blockDim.x + threadIdx.x;
	if (x >= W - r) return;
	const float* __restrict__ hXX = P.hXX + (long long)blockIdx.y * P.hImageStride;
	const float* __restrict__ hXY = P.hXY + (long long)blockIdx.y * P.hImageStride;
	const float* __restrict__ hYY = P.hYY + (long long)blockIdx.y * P.hImageStride;
	float* __restrict__ inten = P.intensity + (long long)blockIdx.y * P.iImageStride;
	float tXX = 0, tXY = 0, tYY = 0;
	for (int k = 0; k < kw; k++) {
		const long long s = (long long)k * W + x;
		tXX += hXX[s]; tXY += hXY[s]; tYY += hYY[s];
	}
	inten[(long long)r * P.iStride + x] = cornerScore(P.kind, P.kappa, tXX, tXY, tYY);
	int y = r + 1;
	for (; y + 8 <= H - r; y += 8) {
		float a[3][8], b[3][8];
#pragma unroll
		for (int k = 0; k < 8; k++) {
			const long long in = (long long)(y + k + r) * W + x, out = in - (long long)kw * W;
			a[0][k] = hXX[out]; a[1][k] = hXY[out]; a[2][k] = hYY[out];
			b[0][k] = hXX[in]; b[1][k] = hXY[in]; b[2][k] = hYY[in];
		}
#pragma unroll
		for (int k = 0; k < 8; k++) {
			tXX = tXX - a[0][k]; tXX += b[0][k];
			tXY = tXY - a[1][k]; tXY += b[1][k];
			tYY = tYY - a[2][k]; tYY += b[2][k];
			inten[(long long)(y + k) * P.iStride + x] = cornerScore(P.kind, P.kappa, tXX, tXY, tYY);
		}
	}
	for (; y < H - r; y++) {
		const long long in = (long long)(y + r) * W + x, out = in - (long long)kw * W;
		tXX = tXX - hXX[out]; tXX += hXX[in];
		tXY = tXY - hXY[out]; tXY += hXY[in];
		tYY = tYY - hYY[out]; tYY += hYY[in];
		inten[(long long)y * P.iStride + x] = cornerScore(P.kind, P.kappa, tXX, tXY, tYY);
	}
}

// intensity must be zero along its border of `radius` pixels: the caller clears the whole image first.  hXX/hXY/hYY: dense width x height
// planes per image, hImageStride floats apart.
int bhip_launch_corner_intensity(bhip_ctx* ctx, int kind, int radius, float kappa, const float* dx, const float* dy, int dStride, int width, int height,
								 float* hXX, float* hXY, float* hYY, float* intensity, int iStride, int batch, long long dImageStride, long long hImageStride,
								 long long iImageStride) {
	if (kind != 0 && kind != 1) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "corner score not supported");
	if (radius < 0 || 2 * radius + 1 > width || 2 * radius + 1 > height) return bhip_fail(ctx, BHIP_ERR_INVALID, "window larger than the image");
	if (batch <= 0) return BHIP_OK;
	CornerParams P{dx, dy, dImageStride, hImageStride, iImageStride, dStride, width, height, radius, hXX, hXY, hYY, intensity, iStride, kind, kappa};
	{
		ProfScope prof(ctx, "k_corner_rows", 4.0 * width * height * 5 * batch);
		hipLaunchKernelGGL(k_corner_rows, dim3((height + 63) / 64, batch), dim3(64), 0, ctx->stream, P);
	}
	{
		ProfScope prof(ctx, "k_corner_cols", 4.0 * width * height * 4 * batch);
		hipLaunchKernelGGL(k_corner_cols, dim3((width - 2 * radius + 255) / 256, batch), dim3(256), 0, ctx->stream, P);
	}
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}


// ---------------- remaining BOverride hooks: 2-D convolution, mean blur, median blur ----------------
// ConvolveImageNoBorder.convolve(Kernel2D_F32)   I:alg/filter/convolve/ConvolveImageNoBorder.java:79-90
//   unrolled widths 3..11: row sums from 0, added in order   I:.../noborder/ConvolveImageUnrolled_SB_F32_F32.java:592-644
//   standard: one running total, row-major                   I:.../noborder/ConvolveImageStandard_SB.java:106-134
#define BHIP_MAX_TAPS2D 441   // 21 x 21
struct Conv2DParams {
	const float* in; float* out;
	int inStride, outStride, width, height, kw, koff, unrolled;
	float k[BHIP_MAX_TAPS2D];
};
__global__ __launch_bounds__(256) void k_conv2d(Conv2DParams P) {
	const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
	const int oL = P.koff, oR = P.kw - P.koff - 1;
	if (x < oL || x >= P.width - oR || y < oL || y >= P.height - oR) return;
	const float* s = P.in + (long long)(y - oL) * P.inStride + (x - oL);
	float total = 0;
	if (P.unrolled) {
		for (int i = 0; i < P.kw; i++) {
			float rowTotal = 0;
			for (int j = 0; j < P.kw; j++) rowTotal += s[(long long)i * P.inStride + j] * P.k[i * P.kw + j];
			total = i == 0 ? rowTotal : total + rowTotal;
		}
	} else {
		int ik = 0;
		for (int i = 0; i < P.kw; i++)
			for (int j = 0; j < P.kw; j++) total += s[(long long)i * P.inStride + j] * P.k[ik++];
	}
	P.out[(long long)y * P.outStride + x] = total;
}
int bhip_launch_conv2d(bhip_ctx* ctx, const float* kernel, int kw, int koff, const float* in, int inStride, int width, int height, float* out, int outStride) {
	if (kw <= 0 || kw * kw > BHIP_MAX_TAPS2D || koff < 0 || koff >= kw) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "kernel width not supported");
	Conv2DParams P;
	P.in = in; P.out = out; P.inStride = inStride; P.outStride = outStride; P.width = width; P.height = height; P.kw = kw; P.koff = koff;
	P.unrolled = (koff == kw / 2 && kw % 2 == 1 && (kw == 3 || kw == 5 || kw == 7 || kw == 9 || kw == 11)) ? 1 : 0;
	for (int i = 0; i < kw * kw; i++) P.k[i] = kernel[i];
	ProfScope prof(ctx, "k_conv2d", 8.0 * width * height);
	hipLaunchKernelGGL(k_conv2d, dim3((width + 255) / 256, height), dim3(256), 0, ctx->stream, P);
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}

// BlurImageOps.mean -> ConvolveImageMean.horizontal / vertical  I:alg/filter/convolve/ConvolveImageMean.java:55-101
//   interior: float running sums, total / divisor   I:alg/filter/convolve/noborder/ImplConvolveMean.java:281-357 (single-threaded order)
//   border: ConvolveNormalized_JustBorder_SB with the table kernel (k_conv mode 3)
// As in the corner intensity, a row (then a column) is one sequential chain; one thread each.
__global__ __launch_bounds__(64) void k_mean_rows(const float* __restrict__ in, int inStride, float* __restrict__ out, int outStride, int width, int height, int radius) {
	const int y = blockIdx.x * blockDim.x + threadIdx.x;
	if (y >= height) return;
	const int kw = 2 * radius + 1;
	const float divisor = (float)kw;
	const float* r = in + (long long)y * inStride;
	float* o = out + (long long)y * outStride;
	float total = 0;
	for (int i = 0; i < kw; i++) total += r[i];
	o[radius] = total / divisor;
	int i = kw;
	for (; i + 16 <= width; i += 16) {
		float a[16], b[16], q[16];
#pragma unroll
		for (int k = 0; k < 16; k++) { a[k] = r[i - kw + k]; b[k] = r[i + k]; }
#pragma unroll
		for (int k = 0; k < 16; k++) { total -= a[k]; total += b[k]; q[k] = total / divisor; }
#pragma unroll
		for (int k = 0; k < 16; k++) o[i - radius + k] = q[k];
	}
	for (; i < width; i++) {
		total -= r[i - kw];
		total += r[i];
		o[i - radius] = total / divisor;
	}
}
__global__ __launch_bounds__(256) void k_mean_cols(const float* __restrict__ in, int inStride, float* __restrict__ out, int outStride, int width, int height, int radius) {
	const int x = blockIdx.x * blockDim.x + threadIdx.x;
	if (x >= width) return;
	const int kw = 2 * radius + 1;
	const float divisor = (float)kw;
	float total = 0;
	for (int k = 0; k < kw; k++) total += in[(long long)k * inStride + x];
	out[(long long)radius * outStride + x] = total / divisor;
	int y = radius + 1;
	for (; y + 8 <= height - radius; y += 8) {
		float a[8], b[8];
#pragma unroll
		for (int k = 0; k < 8; k++) { a[k] = in[(long long)(y + k + radius - kw) * inStride + x]; b[k] = in[(long long)(y + k + radius) * inStride + x]; }
#pragma unroll
		for (int k = 0; k < 8; k++) {
			total = total - a[k];
			total += b[k];
			out[(long long)(y + k) * outStride + x] = total / divisor;
		}
	}
	for (; y < height - radius; y++) {
		total = total - in[(long long)(y + radius - kw) * inStride + x];
		total += in[(long long)(y + radius) * inStride + x];
		out[(long long)y * outStride + x] = total / divisor;
	}
}
// one direction of the mean blur: dense in / out of the same shape
int bhip_launch_mean(bhip_ctx* ctx, bool vertical, const float* in, float* out, int width, int height, int radius) {
	const int kw = 2 * radius + 1;
	if (kw > BHIP_MAX_TAPS) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "mean radius not supported");
	float ker[BHIP_MAX_TAPS];
	const float val = 1.0f / (float)kw;   // FactoryKernel.table1D_F32(radius, true)
	for (int i = 0; i < kw; i++) ker[i] = val;
	const int extent = vertical ? height : width;
	if (kw > extent) return bhip_launch_conv(ctx, vertical, true, ker, kw, radius, in, width, width, height, out, width);   // ConvolveImageNormalized
	// border first (k_conv mode 3), then the interior chains
	{
		ConvParams P;
		P.in = in; P.out = out; P.inStride = width; P.outStride = width; P.width = width; P.height = height; P.kw = kw; P.koff = radius;
		P.inImageStride = 0; P.outImageStride = 0; P.borderOnly = 0;
		for (int i = 0; i < kw; i++) P.k[i] = ker[i];
		P.unrolled = 0; P.mode = 3;
		dim3 grid((width + 255) / 256, height);
		if (vertical) hipLaunchKernelGGL(k_conv<true>, grid, dim3(256), 0, ctx->stream, P);
		else hipLaunchKernelGGL(k_conv<false>, grid, dim3(256), 0, ctx->stream, P);
	}
	ProfScope prof(ctx, vertical ? "k_mean_cols" : "k_mean_rows", 8.0 * width * height);
	if (vertical) hipLaunchKernelGGL(k_mean_cols, dim3((width + 255) / 256), dim3(256), 0, ctx->stream, in, width, out, width, width, height, radius);
	else hipLaunchKernelGGL(k_mean_rows, dim3((height + 63) / 64), dim3(64), 0, ctx->stream, in, width, out, width, width, height, radius);
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}

// BlurImageOps.median(GrayF32) -> ImplMedianSortNaive.process  I:alg/filter/blur/impl/ImplMedianSortNaive.java:97-135: the (count/2)-th order
// statistic of the window clipped to the image.  16x16 pixel tiles staged in LDS with their halo; each thread finds the window
// element v with #(w < v) <= k < #(w <= v) -- comparisons only, so the result is the reference's value exactly.
#define MED_T 16
__global__ __launch_bounds__(MED_T * MED_T) void k_median(const float* __restrict__ in, int inStride, float* __restrict__ out, int outStride, int width, int height,
															int radius) {
	extern __shared__ float tile[];
	const int TW = MED_T + 2 * radius;
	const int x0 = blockIdx.x * MED_T - radius, y0 = blockIdx.y * MED_T - radius;
	for (int i = threadIdx.y * MED_T + threadIdx.x; i < TW * TW; i += MED_T * MED_T) {
		const int ty = i / TW, tx = i - ty * TW;
		const int gx = x0 + tx, gy = y0 + ty;
		tile[i] = (gx >= 0 && gx < width && gy >= 0 && gy < height) ? in[(long long)gy * inStride + gx] : 0.0f;
	}
	__syncthreads();
	const int x = blockIdx.x * MED_T + threadIdx.x, y = blockIdx.y * MED_T + threadIdx.y;
	if (x >= width || y >= height) return;
	const int minI = max(0, y - radius) - y0, maxI = min(height, y + radius + 1) - y0;   // tile coordinates
	const int minJ = max(0, x - radius) - x0, maxJ = min(width, x + radius + 1) - x0;
	const int count = (maxI - minI) * (maxJ - minJ), k = count / 2;
	float result = 0;
	for (int i = minI; i < maxI; i++)
		for (int j = minJ; j < maxJ; j++) {
			const float v = tile[i * TW + j];
			int less = 0, leq = 0;
			for (int a = minI; a < maxI; a++)
				for (int b = minJ; b < maxJ; b++) {
					const float w = tile[a * TW + b];
					less += w < v ? 1 : 0;
					leq += w <= v ? 1 : 0;
				}
			if (less <= k && k < leq) result = v;
		}
	out[(long long)y * outStride + x] = result;
}
int bhip_launch_median(bhip_ctx* ctx, const float* in, int inStride, float* out, int outStride, int width, int height, int radius) {
	if (radius > 8) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "median radius > 8 is not supported on the GPU");
	const int TW = MED_T + 2 * radius;
	ProfScope prof(ctx, "k_median", 8.0 * width * height);
	hipLaunchKernelGGL(k_median, dim3((width + MED_T - 1) / MED_T, (height + MED_T - 1) / MED_T), dim3(MED_T, MED_T), (size_t)TW * TW * 4, ctx->stream, in, inStride, out,
					   outStride, width, height, radius);
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}


// ---------------- integral image of an 8-bit image ----------------
// IntegralImageOps.transform(GrayU8, GrayS32)   I:alg/transform/ii/impl/ImplIntegralImageOps.java:94-118.  Integer sums are exact in any
// order, so the row pass is a wave-parallel scan (one wave per row, 64 pixels per step with a carry) and the column pass one thread per
// column.  Bound: HBM (P read as bytes, 4P written, then 8P for the column pass).
__global__ __launch_bounds__(256) void k_integral_rows_u8(const unsigned char* __restrict__ in, long long inImageStride, int inStride, int* __restrict__ out,
														  long long outImageStride, int outStride, int width, int height) {
	const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
	if (row >= height) return;
	const unsigned char* r = in + (long long)blockIdx.y * inImageStride + (long long)row * inStride;
	int* o = out + (long long)blockIdx.y * outImageStride + (long long)row * outStride;
	int carry = 0;
	for (int x0 = 0; x0 < width; x0 += 64) {
		const int x = x0 + lane;
		int v = x < width ? (int)r[x] : 0;
#pragma unroll
		for (int d = 1; d < 64; d <<= 1) {
			const int t = __shfl_up(v, d, 64);
			if (lane >= d) v += t;
		}
		if (x < width) o[x] = carry + v;
		carry += __shfl(v, 63, 64);
	}
}
__global__ __launch_bounds__(256) void k_integral_cols_s32(int* __restrict__ io, long long imageStride, int stride, int width, int height) {
	const int x = blockIdx.x * blockDim.x + threadIdx.x;
	if (x >= width) return;
	int* p = io + (long long)blockIdx.y * imageStride;
	int total = 0;
	int y = 0;
	for (; y + 8 <= height; y += 8) {   // eight independent loads in flight, then the carry chain
		int v[8];
#pragma unroll
		for (int k = 0; k < 8; k++) v[k] = p[(long long)(y + k) * stride + x];
#pragma unroll
		for (int k = 0; k < 8; k++) { total += v[k]; p[(long long)(y + k) * stride + x] = total; }
	}
	for (; y < height; y++) {
		total += p[(long long)y * stride + x];
		p[(long long)y * stride + x] = total;
	}
}
int bhip_launch_integral_u8(bhip_ctx* ctx, const unsigned char* in, long long inImageStride, int inStride, int* out, long long outImageStride, int outStride,
							int width, int height, int batch) {
	if (width <= 0 || height <= 0 || batch <= 0) return BHIP_OK;
	{
		ProfScope prof(ctx, "k_integral_rows_u8", 5.0 * width * height * batch);
		hipLaunchKernelGGL(k_integral_rows_u8, dim3((height + 3) / 4, batch), dim3(256), 0, ctx->stream, in, inImageStride, inStride, out, outImageStride, outStride,
						   width, height);
	}
	{
		ProfScope prof(ctx, "k_integral_cols_s32", 8.0 * width * height * batch);
		hipLaunchKernelGGL(k_integral_cols_s32, dim3((width + 255) / 256, batch), dim3(256), 0, ctx->stream, out, outImageStride, outStride, width, height);
	}
	BHIP_HIP(ctx, hipGetLastError());
	return BHIP_OK;
}
