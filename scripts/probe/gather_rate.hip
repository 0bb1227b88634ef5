// Probe (not product): what does one wave-wide gather of integral-image taps cost on gfx950, as a function of the sample spacing?
// Mimics k_describe's descriptor sampling (8x8 sample blocks of a rotated 24x24 grid, 12 taps per sample in 10 accesses) and the
// orientation sampling (axis-aligned 17x17 grid), with no other work, so the texture-addresser / L1 cost per instruction can be read off.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/probe/gather_rate scripts/probe/gather_rate.hip
//   scripts/probe/gather_rate            (prints a table)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

struct __attribute__((packed, aligned(4))) F2 { float x, y; };

__device__ __forceinline__ unsigned hashu(unsigned x) {
	x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
	return x;
}

// MODE 0: descriptor pattern, 10 accesses (8 x 4 B + 2 x 8 B)      MODE 1: descriptor pattern, 12 x 4 B
// MODE 2: orientation pattern (axis aligned, lane = sample index), 10 accesses
// MODE 3: descriptor pattern from LDS (random ds_read_b32 in a per-wave region), 12 reads
// MODE 4: descriptor pattern, 1 access per sample (p0 only) -- per-instruction cost without same-line reuse between taps
// cache-policy variants of a 4-byte global load (MODE 5: nt, 6: sc0 sc1, 7: sc1, 8: sc0)
template <int POL>
__device__ __forceinline__ float loadPol(const float* p) {
	float v;
	if (POL == 5) asm volatile("global_load_dword %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
	else if (POL == 6) asm volatile("global_load_dword %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
	else if (POL == 7) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
	else asm volatile("global_load_dword %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
	return v;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_gather(const float* __restrict__ img, int W, int H, int pitch, long long imgStride, int nImg, float s, int kpPerWave,
												 float* __restrict__ out) {
	extern __shared__ float lds[];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int xcd = blockIdx.x & 7;
	const int inX = blockIdx.x >> 3;
	const int perX = gridDim.x >> 3;
	const float* d = img + (long long)(xcd % nImg) * imgStride;
	float acc = 0.f;
	float* myLds = lds + wave * 2560;
	if (MODE == 3) {
		for (int i = lane; i < 2560; i += 64) myLds[i] = (float)i;
		__builtin_amdgcn_s_waitcnt(0);
	}
	for (int k = 0; k < kpPerWave; k++) {
		const unsigned h = hashu((blockIdx.x * 4 + wave) * 131u + k);
		const float margin = 20.f * s + 8.f;
		// waves in flight together sit in the same horizontal band of the frame (the product orders key points by coarse tile)
		const float bandY = margin + (H - 2 * margin) * ((inX * kpPerWave + k) / (float)(perX * kpPerWave));
		const float cx = margin + (W - 2 * margin) * ((h & 0xffff) / 65536.f);
		const float cy = fminf(bandY + ((h >> 16) & 63), H - margin);
		const float ang = ((h >> 22) & 1023) * (6.2831853f / 1024.f);
		const float c = cosf(ang), sn = sinf(ang);
		if (MODE == 2) {
			const int r = ((int)(6.f * s + 0.5f)) / 2;
			const float period = 0.65f * s;
			for (int e = 0; e < 5; e++) {
				const int idx = lane + 64 * e;
				const int sy = idx / 17, sx = idx - sy * 17;
				const bool on = idx < 289;
				const int x = on ? (int)(cx - 8 * period + sx * period) : (int)cx;
				const int y = on ? (int)(cy - 8 * period + sy * period) : (int)cy;
				const unsigned s1 = (unsigned)(y - r - 1) * pitch + (x - r - 1);
				const unsigned s2 = s1 + r * pitch, s3 = s2 + pitch, s4 = s3 + r * pitch;
				const unsigned w = 2 * r + 1;
				const float p0 = d[s1], p3 = d[s1 + w], p11 = d[s2], p4 = d[s2 + w], p10 = d[s3], p5 = d[s3 + w], p9 = d[s4], p6 = d[s4 + w];
				const F2 a = *(const F2*)(d + s1 + r), b = *(const F2*)(d + s4 + r);
				acc += (p6 - b.y - p3 + a.y) - (b.x - p9 - a.x + p0) + (p6 - p9 - p5 + p10) - (p4 - p11 - p3 + p0);
			}
		} else if (MODE == 9 || MODE == 10) {
			// quad-per-sample mapping: 16 samples per instruction, the four lanes of a quad take the four taps of one image row of their
			// sample (row a, rows c1+c2, row b: three instructions per 16 samples, 108 per key point); MODE 10: 4x4 sample groups
			const int r = max(1, ((int)(3.f * s + 0.5f)) / 2);
			const int q = lane >> 2, k = lane & 3;
			const unsigned colOff = k == 0 ? 0u : k == 1 ? (unsigned)r : k == 2 ? (unsigned)r + 1u : 2u * r + 1u;
			const unsigned offB = (unsigned)(k < 2 ? r : r + 1) * pitch + ((k & 1) ? 2u * r + 1u : 0u);
			const unsigned offC = (unsigned)(2 * r + 1) * pitch + colOff;
#pragma unroll 1
			for (int step = 0; step < 12; step++) {
				float A[3], B[3], C[3];
#pragma unroll
				for (int u = 0; u < 3; u++) {
					int ix, iy;
					if (MODE == 9) { ix = 8 * u + (q & 7); iy = 2 * step + (q >> 3); }
					else { const int g = step * 3 + u; ix = 4 * (g % 6) + (q & 3); iy = 4 * (g / 6) + (q >> 2); }
					const float rY = (iy - 12) * s, rX = (ix - 12) * s;
					const int x = (int)(cx + c * rX - sn * rY);
					const int y = (int)(cy + sn * rX + c * rY);
					const unsigned s1 = (unsigned)(y - r - 1) * pitch + (x - r - 1);
					A[u] = d[s1 + colOff]; B[u] = d[s1 + offB]; C[u] = d[s1 + offC];
				}
#pragma unroll
				for (int u = 0; u < 3; u++) acc += (A[u] - B[u]) + C[u];
			}
		} else {
			const int r = max(1, ((int)(3.f * s + 0.5f)) / 2);
			const int ly = lane >> 3, lx = lane & 7;
#pragma unroll 3
			for (int b = 0; b < 9; b++) {
				const int by = b / 3, bx = b - by * 3;
				const float rY = (8 * by + ly - 12) * s, rX = (8 * bx + lx - 12) * s;
				const int x = (int)(cx + c * rX - sn * rY);
				const int y = (int)(cy + sn * rX + c * rY);
				if (MODE == 3) {
					// same count of reads, from LDS: pseudo-random addresses in the wave's 10 KB
					unsigned a0 = hashu(x * 7919u + y) % 2500u;
					float v = 0;
#pragma unroll
					for (int t = 0; t < 12; t++) v += myLds[a0 + t * 5];
					acc += v;
					continue;
				}
				const unsigned s1 = (unsigned)(y - r - 1) * pitch + (x - r - 1);
				const unsigned s2 = s1 + r * pitch, s3 = s2 + pitch, s4 = s3 + r * pitch;
				const unsigned w = 2 * r + 1;
				if (MODE == 4) { acc += d[s1]; continue; }
				if (MODE >= 5) {
					// the twelve taps as 4-byte loads with a cache policy; waited for once per sample block
					float t[12];
					const unsigned o[12] = {s1, s1 + r, s1 + r + 1, s1 + w, s2, s2 + w, s3, s3 + w, s4, s4 + r, s4 + r + 1, s4 + w};
#pragma unroll
					for (int q = 0; q < 12; q++) t[q] = loadPol<MODE>(d + o[q]);
					asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
					for (int q = 0; q < 12; q++) acc += t[q];
					continue;
				}
				const float p0 = d[s1], p3 = d[s1 + w], p11 = d[s2], p4 = d[s2 + w], p10 = d[s3], p5 = d[s3 + w], p9 = d[s4], p6 = d[s4 + w];
				float p1, p2, p8, p7;
				if (MODE == 0) {
					const F2 a = *(const F2*)(d + s1 + r), bb = *(const F2*)(d + s4 + r);
					p1 = a.x; p2 = a.y; p8 = bb.x; p7 = bb.y;
				} else {
					p1 = d[s1 + r]; p2 = d[s1 + r + 1]; p8 = d[s4 + r]; p7 = d[s4 + r + 1];
				}
				acc += (p6 - p7 - p3 + p2) - (p8 - p9 - p1 + p0) + (p6 - p9 - p5 + p10) - (p4 - p11 - p3 + p0);
			}
		}
	}
	out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
double run(const float* img, int W, int H, long long imgStride, int nImg, float s, int blocks, int kpPerWave, size_t ldsBytes, float* out) {
	hipEvent_t a, b;
	CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
	if (ldsBytes > 65536) CHECK(hipFuncSetAttribute((const void*)k_gather<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
	hipLaunchKernelGGL((k_gather<MODE>), dim3(blocks), dim3(256), ldsBytes, 0, img, W, H, W, imgStride, nImg, s, kpPerWave, out);
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(a));
	hipLaunchKernelGGL((k_gather<MODE>), dim3(blocks), dim3(256), ldsBytes, 0, img, W, H, W, imgStride, nImg, s, kpPerWave, out);
	CHECK(hipEventRecord(b));
	CHECK(hipEventSynchronize(b));
	float ms = 0;
	CHECK(hipEventElapsedTime(&ms, a, b));
	return ms;
}

int main(int argc, char** argv) {
	const int W = 1920, H = 1080, nImg = 8;
	const long long imgStride = (long long)W * H;
	float* img; float* out;
	CHECK(hipMalloc(&img, imgStride * nImg * 4));
	std::vector<float> h(imgStride);
	for (long long i = 0; i < imgStride; i++) h[i] = (float)(i % 977);
	for (int i = 0; i < nImg; i++) CHECK(hipMemcpy(img + i * imgStride, h.data(), imgStride * 4, hipMemcpyHostToDevice));
	const int blocks = 256 * 4 * 8, kpPerWave = 4;   // 32k waves x 4 key points
	CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
	const double kps = (double)blocks * 4 * kpPerWave;
	const double clk = 2.4e9, cus = 256;
	printf("key points per launch %.0f; CU-cycles per key point = ms * 1e-3 * 2.4e9 * 256 / kps\n", kps);
	const float scales[] = {2.f, 3.f, 4.f, 5.4f, 8.f, 12.f, 18.f};
	printf("cache policies, 12 x 4-byte taps per sample, 40 KB LDS per workgroup: ms per launch\n%6s %10s %10s %10s %10s %10s\n", "scale", "default", "nt", "sc0 sc1", "sc1", "sc0");
	for (float s : scales) {
		const double a = run<1>(img, W, H, imgStride, nImg, s, blocks, kpPerWave, 40960, out);
		const double b5 = run<5>(img, W, H, imgStride, nImg, s, blocks, kpPerWave, 40960, out);
		const double b6 = run<6>(img, W, H, imgStride, nImg, s, blocks, kpPerWave, 40960, out);
		const double b7 = run<7>(img, W, H, imgStride, nImg, s, blocks, kpPerWave, 40960, out);
		const double b8 = run<8>(img, W, H, imgStride, nImg, s, blocks, kpPerWave, 40960, out);
		printf("%6.1f %10.3f %10.3f %10.3f %10.3f %10.3f\n", s, a, b5, b6, b7, b8);
	}
	if (argc > 1) return 0;
	const size_t ldsOpts[] = {40960, 20480, 10240};
	for (size_t lb : ldsOpts) {
		printf("--- dynamic LDS per workgroup %zu B (%d workgroups / CU by LDS)\n", lb, (int)(163840 / lb));
		printf("%6s %26s %26s %26s %26s %26s\n", "scale", "desc 10acc ms (cyc/kp, /instr)", "desc 12acc", "ori 10acc x5", "desc LDS 12rd", "desc 1acc");
		for (float s : scales) {
			const double m0 = run<0>(img, W, H, imgStride, nImg, s, blocks, kpPerWave, lb, out);
			const double m1 = run<1>(img, W, H, imgStride, nImg, s, blocks, kpPerWave, lb, out);
			const double m2 = run<2>(img, W, H, imgStride, nImg, s, blocks, kpPerWave, lb, out);
			const double m3 = run<3>(img, W, H, imgStride, nImg, s, blocks, kpPerWave, lb, out);
			const double m4 = run<4>(img, W, H, imgStride, nImg, s, blocks, kpPerWave, lb, out);
			const double m9 = run<9>(img, W, H, imgStride, nImg, s, blocks, kpPerWave, lb, out);
			const double m10 = run<10>(img, W, H, imgStride, nImg, s, blocks, kpPerWave, lb, out);
			auto cyc = [&](double ms) { return ms * 1e-3 * clk * cus / kps; };
			printf("%6.1f   %7.3f (%6.0f, %5.1f)   %7.3f (%6.0f, %5.1f)   %7.3f (%6.0f, %5.1f)   %7.3f (%6.0f, %5.1f)   %7.3f (%6.0f, %5.1f)\n", s,
				   m0, cyc(m0), cyc(m0) / 90, m1, cyc(m1), cyc(m1) / 108, m2, cyc(m2), cyc(m2) / 50, m3, cyc(m3), cyc(m3) / 108, m4, cyc(m4), cyc(m4) / 9);
			printf("         quad-per-sample 8x2: %7.3f (%6.0f, %5.1f)   4x4: %7.3f (%6.0f, %5.1f)\n", m9, cyc(m9), cyc(m9) / 108, m10, cyc(m10), cyc(m10) / 108);
		}
	}
	return 0;
}
