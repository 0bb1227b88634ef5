/* Direct C caller of the drop-in boundary (include/boofhip.h), exactly the sequence a JNI shim performs for
 *   DetectDescribePoint.detect(GrayF32)    -> bhip_surf_create / bhip_surf_detect_f32 / bhip_surf_count / bhip_surf_fetch
 *   AssociateDescription.associate()       -> bhip_assoc_l2_f64
 * (DetectDescribePoint.java:32-46, AssociateDescription.java:42-61).  Built with plain gcc against libboofhip.so by tests/test_cabi_direct.py,
 * which supplies two raw float32 images and compares the dumped results with the CPU oracle.  No Python, ctypes or torch on this path.
 *
 * usage: cabi_direct W H imgA.f32 imgB.f32 out.bin
 * out.bin: int32 nA, nB; then for A and B: xy_scale[3n] f64, angle[n] f64, white[n] u8, desc[64n] f64; then pairs[nA] i32, fit[nA] f64 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "boofhip.h"

#define CHECK(call)                                                                                       \
	do {                                                                                                  \
		int st_ = (call);                                                                                 \
		if (st_ != BHIP_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, st_, bhip_last_error(ctx)); return 2; } \
	} while (0)

static float* readImage(const char* path, size_t n) {
	FILE* f = fopen(path, "rb");
	if (!f) return NULL;
	float* p = (float*)malloc(n * sizeof(float));
	size_t got = fread(p, sizeof(float), n, f);
	fclose(f);
	if (got != n) { free(p); return NULL; }
	return p;
}

int main(int argc, char** argv) {
	if (argc != 6) { fprintf(stderr, "usage: %s W H imgA.f32 imgB.f32 out.bin\n", argv[0]); return 1; }
	const int W = atoi(argv[1]), H = atoi(argv[2]);
	float* img[2] = {readImage(argv[3], (size_t)W * H), readImage(argv[4], (size_t)W * H)};
	if (!img[0] || !img[1]) { fprintf(stderr, "cannot read the images\n"); return 1; }
	bhip_ctx* ctx = NULL;
	if (bhip_ctx_create(0, &ctx) != BHIP_OK) { fprintf(stderr, "bhip_ctx_create failed: no usable GPU (there is no CPU fallback)\n"); return 3; }
	bhip_surf* surf = NULL;
	CHECK(bhip_surf_create(ctx, NULL, NULL, NULL, 1, &surf));   /* FactoryDetectDescribe.surfStable(null, null, null, GrayF32.class) */
	FILE* out = fopen(argv[5], "wb");
	if (!out) return 1;
	int n[2] = {0, 0};
	double* desc[2] = {NULL, NULL};
	long header = ftell(out);
	fwrite(n, sizeof(int), 2, out);
	for (int k = 0; k < 2; k++) {
		const float* one[1] = {img[k]};
		CHECK(bhip_surf_detect_f32(surf, one, NULL, NULL, W, H, 1));          /* detect(input) */
		CHECK(bhip_surf_count(surf, 0, &n[k]));                               /* getNumberOfFeatures() */
		const size_t m = (size_t)(n[k] > 0 ? n[k] : 1);
		double* xys = (double*)malloc(m * 3 * sizeof(double));
		double* ang = (double*)malloc(m * sizeof(double));
		uint8_t* white = (uint8_t*)malloc(m);
		desc[k] = (double*)malloc(m * 64 * sizeof(double));
		CHECK(bhip_surf_fetch(surf, 0, xys, ang, white, desc[k]));            /* getLocation / getRadius / getOrientation / getDescription */
		fwrite(xys, sizeof(double), (size_t)n[k] * 3, out);
		fwrite(ang, sizeof(double), (size_t)n[k], out);
		fwrite(white, 1, (size_t)n[k], out);
		fwrite(desc[k], sizeof(double), (size_t)n[k] * 64, out);
		free(xys); free(ang); free(white);
	}
	int* pairs = (int*)malloc((size_t)(n[0] > 0 ? n[0] : 1) * sizeof(int));
	double* fit = (double*)malloc((size_t)(n[0] > 0 ? n[0] : 1) * sizeof(double));
	/* FactoryAssociation.greedy(new ScoreAssociateEuclideanSq_F64(), Double.MAX_VALUE, true): setSource(A), setDestination(B), associate() */
	CHECK(bhip_assoc_l2_f64(ctx, desc[0], n[0], desc[1], n[1], 64, 1.7976931348623157e308, 1, 0, pairs, fit));
	fwrite(pairs, sizeof(int), (size_t)n[0], out);
	fwrite(fit, sizeof(double), (size_t)n[0], out);
	fseek(out, header, SEEK_SET);
	fwrite(n, sizeof(int), 2, out);
	fclose(out);
	CHECK(bhip_surf_destroy(surf));
	CHECK(bhip_ctx_destroy(ctx));
	printf("cabi_direct ok: %d / %d key points\n", n[0], n[1]);
	return 0;
}
