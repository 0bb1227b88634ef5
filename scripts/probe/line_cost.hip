// Probe (not product): cost of one wave-wide 4-byte gather on gfx950 as a function of the number of distinct 128-byte lines it touches
// and of where those lines live (L1-resident working set vs L2).  Lanes are grouped G consecutive lanes per line (G = 64, 32, ..., 1).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/probe/line_cost scripts/probe/line_cost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ unsigned hashu(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// wsLines: working set in lines per workgroup-region; each CU's waves draw lines from a region of that many lines
// SPREAD: 0 = the G lanes of a group read consecutive dwords of the line, 1 = all read the same dword
template <int G, int WIDTH>
__global__ __launch_bounds__(256) void k_lines(const float* __restrict__ img, unsigned wsLines, int iters, float* __restrict__ out, unsigned nRegions) {
	const int lane = threadIdx.x & 63;
	const unsigned grp = lane / G, within = lane % G;
	const unsigned regionBase = (blockIdx.x % nRegions) * wsLines;   // lines
	float acc = 0.f;
	unsigned h = hashu(blockIdx.x * 256u + (threadIdx.x >> 6) * 64u);
#pragma unroll 1
	for (int i = 0; i < iters; i += 8) {
		float v[8][WIDTH];
#pragma unroll
		for (int u = 0; u < 8; u++) {
			h = h * 1664525u + 1013904223u;
			const unsigned line = hashu(h + grp * 977u) % wsLines;
			const float* p = img + (size_t)(regionBase + line) * 32u + (within * WIDTH) % 32u;
			if (WIDTH == 1) v[u][0] = *p;
			else if (WIDTH == 2) { const float2 t = *(const float2*)p; v[u][0] = t.x; v[u][1] = t.y; }
			else { const float4 t = *(const float4*)p; v[u][0] = t.x; v[u][1] = t.y; v[u][2] = t.z; v[u][3] = t.w; }
		}
#pragma unroll
		for (int u = 0; u < 8; u++)
#pragma unroll
			for (int w = 0; w < WIDTH; w++) acc += v[u][w];
	}
	out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int G, int WIDTH>
double run(const float* img, unsigned wsLines, float* out, unsigned nRegions = 1024) {
	const int blocks = 256 * 4 * 4, iters = 512;
	hipEvent_t a, b;
	CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
	hipLaunchKernelGGL((k_lines<G, WIDTH>), dim3(blocks), dim3(256), 0, 0, img, wsLines, iters, out, nRegions);
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(a));
	hipLaunchKernelGGL((k_lines<G, WIDTH>), dim3(blocks), dim3(256), 0, 0, img, wsLines, iters, out, nRegions);
	CHECK(hipEventRecord(b));
	CHECK(hipEventSynchronize(b));
	float ms = 0;
	CHECK(hipEventElapsedTime(&ms, a, b));
	// CU-cycles per wave instruction: ms * clk * CUs / (waves * iters)
	return ms * 1e-3 * 2.4e9 * 256 / ((double)blocks * 4 * iters);
}

int main() {
	const size_t lines = (size_t)1024 * 65536;   // 8 GB would be too much: regions wrap (blockIdx % 1024) * ws, ws <= 65536 lines -> 8.6 GB; cap below
	float* img; float* out;
	const size_t maxWs = 16384;   // lines per region (2 MB); 1024 regions = 2 GB
	CHECK(hipMalloc(&img, (size_t)1024 * maxWs * 128));
	CHECK(hipMemset(img, 0, (size_t)1024 * maxWs * 128));
	CHECK(hipMalloc(&out, (size_t)256 * 16 * 256 * 4));
	(void)lines;
	const unsigned wss[] = {64, 192, 1024, 16384};   // 8 KB, 24 KB (L1-resident per CU: 4 blocks/CU share... each block has its own region), 128 KB, 2 MB
	printf("CU-cycles per wave-level load instruction; columns = distinct lines per instruction\n");
	for (unsigned ws : wss) {
		printf("working set %u lines (%u KB) per workgroup\n", ws, ws / 8);
		printf("  width      64 lines   32 lines   16 lines    8 lines    4 lines    2 lines    1 line\n");
		printf("  dword    %9.1f  %9.1f  %9.1f  %9.1f  %9.1f  %9.1f  %9.1f\n", run<1, 1>(img, ws, out), run<2, 1>(img, ws, out), run<4, 1>(img, ws, out), run<8, 1>(img, ws, out),
			   run<16, 1>(img, ws, out), run<32, 1>(img, ws, out), run<64, 1>(img, ws, out));
		printf("  dwordx2  %9.1f  %9.1f  %9.1f  %9.1f  %9.1f  %9.1f  %9.1f\n", run<1, 2>(img, ws, out), run<2, 2>(img, ws, out), run<4, 2>(img, ws, out), run<8, 2>(img, ws, out),
			   run<16, 2>(img, ws, out), run<32, 2>(img, ws, out), run<64, 2>(img, ws, out));
		printf("  dwordx4  %9.1f  %9.1f  %9.1f  %9.1f  %9.1f  %9.1f  %9.1f\n", run<1, 4>(img, ws, out), run<2, 4>(img, ws, out), run<4, 4>(img, ws, out), run<8, 4>(img, ws, out),
			   run<16, 4>(img, ws, out), run<32, 4>(img, ws, out), run<64, 4>(img, ws, out));
	}
	// L1 misses that hit the XCD's L2: every workgroup draws from the same few regions (2 MB in all, resident in each XCD's 4 MB L2)
	printf("L2-resident: 16 regions x 1024 lines shared by all workgroups (2 MB)\n");
	printf("  dword    %9.1f  %9.1f  %9.1f  %9.1f  %9.1f  %9.1f  %9.1f\n", run<1, 1>(img, 1024, out, 16), run<2, 1>(img, 1024, out, 16), run<4, 1>(img, 1024, out, 16), run<8, 1>(img, 1024, out, 16),
		   run<16, 1>(img, 1024, out, 16), run<32, 1>(img, 1024, out, 16), run<64, 1>(img, 1024, out, 16));
	printf("  dwordx4  %9.1f  %9.1f  %9.1f  %9.1f  %9.1f  %9.1f  %9.1f\n", run<1, 4>(img, 1024, out, 16), run<2, 4>(img, 1024, out, 16), run<4, 4>(img, 1024, out, 16), run<8, 4>(img, 1024, out, 16),
		   run<16, 4>(img, 1024, out, 16), run<32, 4>(img, 1024, out, 16), run<64, 4>(img, 1024, out, 16));
	printf("L2-resident: 4 regions x 1024 lines (512 KB)\n");
	printf("  dword    %9.1f  %9.1f  %9.1f  %9.1f  %9.1f  %9.1f  %9.1f\n", run<1, 1>(img, 1024, out, 4), run<2, 1>(img, 1024, out, 4), run<4, 1>(img, 1024, out, 4), run<8, 1>(img, 1024, out, 4),
		   run<16, 1>(img, 1024, out, 4), run<32, 1>(img, 1024, out, 4), run<64, 1>(img, 1024, out, 4));
	return 0;
}
