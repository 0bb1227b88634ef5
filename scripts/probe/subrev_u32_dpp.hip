#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int P0, int P1, int P2, int P3>
__device__ __forceinline__ int qp(int v) { return __builtin_amdgcn_update_dpp(0, v, P0 | (P1 << 2) | (P2 << 4) | (P3 << 6), 0xf, 0xf, true); }
__global__ void k(const int* in, int* out) {
	const int t = threadIdx.x;
	const int x = in[t];
	// variant 0: folded form
	out[t] = qp<1, 1, 3, 3>(x) - qp<0, 0, 2, 2>(x);
	// variant 1: both operands as explicit moves
	int a = qp<1, 1, 3, 3>(x), b = qp<0, 0, 2, 2>(x);
	asm volatile("" : "+v"(a), "+v"(b));
	out[64 + t] = a - b;
	// variant 2: explicit asm of the subrev form
	int r;
	asm volatile("s_nop 4\n\tv_subrev_u32_dpp %0, %1, %2 quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 4" : "=v"(r) : "v"(x), "v"(a));
	out[128 + t] = r;
	// variant 3: v_sub_u32_dpp a' - b with dpp on the minuend
	int r2;
	asm volatile("s_nop 4\n\tv_sub_u32_dpp %0, %1, %2 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 4" : "=v"(r2) : "v"(x), "v"(b));
	out[192 + t] = r2;
	out[256 + t] = a; out[320 + t] = b;
}
int main() {
	std::vector<int> h(64), o(384);
	for (int i = 0; i < 64; i++) h[i] = 1000 * i + 7 * (i % 5);
	int *d, *od; (void)hipMalloc(&d, 256); (void)hipMalloc(&od, 384 * 4);
	(void)hipMemcpy(d, h.data(), 256, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, od);
	(void)hipMemcpy(o.data(), od, 384 * 4, hipMemcpyDeviceToHost);
	for (int i = 0; i < 8; i++) {
		const int q = i & ~3; const int want = h[q + ((i & 2) ? 3 : 1)] - h[q + ((i & 2) ? 2 : 0)];
		printf("lane %d x %d want %d | folded %d  moves %d  subrev_asm %d  sub_asm %d | a %d b %d\n", i, h[i], want, o[i], o[64 + i], o[128 + i], o[192 + i], o[256 + i], o[320 + i]);
	}
	return 0;
}
