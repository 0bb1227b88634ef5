// Probe: operand lane map of v_mfma_i32_32x32x32_i8 on gfx950 (the guide says: check other dtypes with exact integer data).
// Candidate 0: lane l holds A[row l&31][k = 16*(l>>5) + j], j = 0..15 (byte j of the 16-byte fragment), B likewise with columns.
// Candidate 1: k = 8*(l>>5) + (j & 7) + 16*(j >> 3)   (two 8-wide halves interleaved, as two stacked 32x32x16 steps)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__global__ void probe(const int8_t* A, const int8_t* B, int* C, int cand) {
	const int l = threadIdx.x, r = l & 31, h = l >> 5;
	int8_t a[16], b[16];
	for (int j = 0; j < 16; j++) {
		const int k = cand == 0 ? 16 * h + j : 8 * h + (j & 7) + 16 * (j >> 3);
		a[j] = A[r * 32 + k];      // A[row][k]
		b[j] = B[k * 32 + r];      // B[k][col]
	}
	v4i av, bv;
	__builtin_memcpy(&av, a, 16);
	__builtin_memcpy(&bv, b, 16);
	v16i acc = {0};
	acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, acc, 0, 0, 0);
	for (int g = 0; g < 16; g++) {
		const int row = (g & 3) + 8 * (g >> 2) + 4 * h, col = r;
		C[row * 32 + col] = acc[g];
	}
}
int main() {
	int8_t hA[1024], hB[1024];
	srand(1);
	for (int i = 0; i < 1024; i++) { hA[i] = rand() % 7 - 3; hB[i] = rand() % 5 - 2; }
	int ref[1024];
	for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { int s = 0; for (int k = 0; k < 32; k++) s += hA[i * 32 + k] * hB[k * 32 + j]; ref[i * 32 + j] = s; }
	int8_t *dA, *dB; int* dC;
	hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 4096);
	hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
	for (int cand = 0; cand < 2; cand++) {
		hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, cand);
		int hC[1024];
		hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost);
		int bad = 0;
		for (int i = 0; i < 1024; i++) bad += hC[i] != ref[i];
		printf("candidate %d: %d mismatches of 1024\n", cand, bad);
	}
	return 0;
}
