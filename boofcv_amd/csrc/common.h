// Internal declarations shared by the HIP translation units of libboofhip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>
#include <new>
#include "../../include/boofhip.h"

#define BHIP_WAVE 64

// Parity cross-check hooks (BHIP_DETECT_UNFUSED, BHIP_ASSOC_EXACT, ...) select an alternative, equally exact execution plan; they are read
// from the environment on every call (no process-wide cache: contexts on different host threads may not share mutable statics).
// Timing-experiment switches (ablation, tile variants, stamps) exist only in -DBHIP_EXPERIMENTS builds (boofcv_amd/build.py --experiments),
// never in the shipped libboofhip.so.
static inline bool bhip_env_flag(const char* name) {
	const char* e = getenv(name);
	return e && e[0] == '1';
}
#ifdef BHIP_EXPERIMENTS
#define BHIP_ABLATE(P, bits) ((P).ablate & (bits))
#else
#define BHIP_ABLATE(P, bits) 0
#endif

// one bracketed kernel launch (profiling mode only)
struct ProfRecord {
	const char* tag;
	double algBytes;     // algorithmic HBM bytes of this launch, 0 when not an HBM-roofline kernel
	double algFlops;     // algorithmic flops / integer ops of this launch, 0 when not a compute-roofline kernel
	hipEvent_t start, stop;
};

struct bhip_ctx {
	int device = 0;
	hipStream_t stream = nullptr;
	bool ownStream = false;
	std::string error;
	// small pinned staging buffer for count read-backs
	int* hostScratch = nullptr;
	// optional per-kernel HIP-event timing on the ctx stream (bhip_profile_*)
	bool profiling = false;
	std::vector<ProfRecord> profRecords;
	std::vector<hipEvent_t> eventPool;
	size_t integralLdsAttr = 0;   // largest dynamic-LDS size k_integral_fused has been configured for on this ctx's device
};

// RAII bracket around one kernel launch: records a start/stop event pair on the ctx stream when profiling is on
struct ProfScope {
	bhip_ctx* ctx;
	int idx = -1;
	ProfScope(bhip_ctx* c, const char* tag, double algBytes = 0, double algFlops = 0);
	~ProfScope();
};

void bhip_profile_release(bhip_ctx* ctx);   // destroys the ctx's profiling events (profile.hip)

static inline int bhip_fail(bhip_ctx* ctx, int code, const std::string& msg) {
	if (ctx) ctx->error = msg;
	return code;
}

#define BHIP_HIP(ctx, expr)                                                                                                      \
	do {                                                                                                                         \
		hipError_t _e = (expr);                                                                                                  \
		if (_e != hipSuccess) return bhip_fail((ctx), _e == hipErrorOutOfMemory ? BHIP_ERR_NOMEM : BHIP_ERR_HIP,                 \
											   std::string(#expr) + ": " + hipGetErrorString(_e));                               \
	} while (0)

#define BHIP_TRY(expr)                 \
	do {                               \
		int _s = (expr);               \
		if (_s != BHIP_OK) return _s;  \
	} while (0)

// grow-only device buffer
struct DevBuf {
	void* p = nullptr;
	size_t cap = 0;
	int reserve(bhip_ctx* ctx, size_t bytes) {
		if (bytes <= cap) return BHIP_OK;
		if (p) { BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream)); BHIP_HIP(ctx, hipFree(p)); p = nullptr; cap = 0; }
		size_t want = bytes + bytes / 8;
		BHIP_HIP(ctx, hipMalloc(&p, want));
		cap = want;
		return BHIP_OK;
	}
	void release() { if (p) { (void)hipFree(p); p = nullptr; cap = 0; } }
	template <class T> T* as() const { return (T*)p; }
};

// work buffers of the MFMA association path (assoc_mfma.hip)
struct AssocMfmaWork {
	DevBuf Fs, Fd, nrmS, nrmD, probs, blocks, keys, thr, best, args, cand, flags;
};

// ---------------- image view passed to kernels ----------------
struct ImgView {
	const float* data;   // base of image 0 (already offset by startIndex)
	long long imageStride;  // floats between images of a batch
	int stride;          // floats between rows
	int width, height;
};
struct ImgViewW {
	float* data;
	long long imageStride;
	int stride;
	int width, height;
};

// ---------------- detector structures ----------------
#define BHIP_MAX_OCTAVES 8
#define BHIP_MAX_LEVELS 8

struct KeyPoint {      // one detected interest point, as stored on the device
	double x, y, scale;
	unsigned int key;    // bit index in the per-image candidate bitmap == rank key (octave, level, blockY, blockX)
	unsigned int pad;
};

// tables for orientation + descriptor, resident in device memory
struct SurfTables {
	// orientation
	int oriStable;           // 1 sliding window, 0 average
	int oriRadius;           // sampleRadius
	int oriWidth;            // 2*radius+1
	int oriKernelWidth;      // ConfigOrientation.sampleWidth
	int oriHasWeights;
	double oriPeriod;        // samplePeriod
	double oriWindow;        // windowSize
	double oriRadiusToScale; // objectRadiusToScale
	const double* oriWeights;  // oriWidth^2
	// descriptor
	int stable;              // 1 DescribePointSurfMod, 0 DescribePointSurf
	int widthLargeGrid, widthSubRegion, widthSample, overLap;
	int dof;
	int radiusDescriptor;
	const double* weightSub;   // (sub+2*overlap)^2        (stable)
	const double* weightGrid;  // largeGrid^2              (stable)
	const double* weightFast;  // (largeGrid*sub)^2        (fast)
};

// host-side Gaussian tables (tables.cpp): product code, independent of oracle/
std::vector<double> bhip_gaussian2d_f64(double sigma, int radius);          // FactoryKernelGaussian.gaussian(2,true,64,sigma,radius)
std::vector<double> bhip_gaussian_width(double sigma, int width);           // FactoryKernelGaussian.gaussianWidth
std::vector<float> bhip_gaussian1d_f32(double sigma, int radius);           // FactoryKernelGaussian.gaussian(Kernel1D_F32,sigma,radius)

// ---------------- kernel launchers (defined in the .hip files) ----------------
int bhip_launch_integral(bhip_ctx* ctx, ImgView in, ImgViewW out, int batch);
// levels a fused octave writes out for the next octave (every second pixel, next octave's [image][slot][h][w] layout)
struct FusedExport {
	int n;
	int level[2];
	float* out;
	int w, h;
	long long imageStride;
	long long slotOffset[2];   // floats from an image's base to the slot's [h][w] plane (the consuming octave's level planes when it is written in place)
};
// where a stand-alone Hessian level comes from: computed, or copied from a level of the same kernel size one octave down
struct HessLevelSource {
	const float* src;        // nullptr: compute
	long long imageStride;   // floats between images at src
	int stride;              // floats between rows at src
	int step;                // 1: src already has this octave's layout, 2: take every second pixel
	int inPlace;             // src IS this level's output plane (the producer wrote it there): only the pixels that must be recomputed are touched
};
int bhip_launch_hessian(bhip_ctx* ctx, ImgView ii, int batch, int skip, int nlevels, const int* sizes, float* intensity, long long levelStride,
						long long imageStrideOut, int outStride, const HessLevelSource* from = nullptr, bool intTaps = false,
						unsigned int skipMask = 0);   // skipMask: levels this launch does not produce

int bhip_launch_copy_images(bhip_ctx* ctx, const float* in, long long inImageStride, int inStride, float* out, long long outImageStride, int outStride,
							int width, int height, int batch);

// describe-kernel options beyond the grey float default (see DescParams): colour SURF bands (nBands > 0), the orientation's object
// radius factor, integer taps (GrayS32 integral images)
struct DescPlanar {
	const float* data;               // band integral images, [image][band][H][W]
	long long imageStride, bandStride;
	int nBands;
	double oriRadiusFactor;
	bool intTaps;
};
struct DetectLevelParams {
	int skip, w, h;              // intensity image size
	int sizeLower, sizeMid, sizeUpper;  // kernel sizes of level-1, level, level+1
	int border;                  // ignoreBorder = size/(2*skip)
	int nbx, nby;                // blocks in the NMS region
	unsigned int bitBase;        // first bit of this (octave,level) in the per-image bitmap
};
int bhip_launch_nms_scalespace(bhip_ctx* ctx, const float* lower, const float* mid, const float* upper, long long imageStride, int stride, int batch,
							   DetectLevelParams p, int radius, float threshold, unsigned int* bitmap, int bitmapWords, KeyPoint* cand, int* candCount,
							   int cap, bool listOnly = false, const ImgView* ii = nullptr, bool intTaps = false);   // lower / upper == nullptr: evaluated on demand from *ii
int bhip_launch_select_nbest(bhip_ctx* ctx, const float* lower, const float* mid, const float* upper, long long imageStride, int stride, int batch,
							 DetectLevelParams p, int radius, int target, const unsigned int* bitmap, const unsigned int* prefix, int bitmapWords,
							 const KeyPoint* nms, int cap, float* keyBuf, int* idxBuf, KeyPoint* out, int* levelStart, int* levelCount, int levelIndex, int nlv);
int bhip_launch_select_nbest_xy(bhip_ctx* ctx, const float* img, int stride, const int16_t* xy, int n, int target, bool positive, float* key, int* idx,
								int16_t* out);
int bhip_launch_compact_levels(bhip_ctx* ctx, const KeyPoint* src, int cap, const int* levelStart, const int* levelCount, int nlv, int batch, KeyPoint* dst,
							   int* totals);
int bhip_launch_rank_scatter(bhip_ctx* ctx, const unsigned int* bitmap, int bitmapWords, unsigned int* wordPrefix, const KeyPoint* cand,
							 const int* candCount, int cap, int batch, KeyPoint* sorted);
int bhip_launch_word_prefix(bhip_ctx* ctx, const unsigned int* bitmap, int bitmapWords, int batch, unsigned int* wordPrefix, int* totals);

int bhip_launch_describe(bhip_ctx* ctx, ImgView ii, const KeyPoint* kps, const int* kpImage /*may be null: use imageOfBlock*/, long long total,
						 const int* imageStart /*batch+1 prefix of counts*/, int batch, SurfTables t, double* angles, double* desc, uint8_t* white);

int bhip_launch_assoc_l2(bhip_ctx* ctx, const double* src, int ns, const double* dst, int nd, int dof, double maxErr, int backwards, int sqrtScore,
						 int* pairs, double* fit, DevBuf& work);
int bhip_launch_assoc_hamming(bhip_ctx* ctx, const int32_t* src, int ns, const int32_t* dst, int nd, int words, double maxErr, int backwards,
							  int* pairs, double* fit, DevBuf& work);

// ---------------- boofcv-ip front end (ip.hip); device images, `batch` of them imageStride floats apart ----------------
int bhip_launch_conv(bhip_ctx* ctx, bool vertical, bool normalized, const float* kernel, int kw, int koff, const float* in, int inStride, int width,
					 int height, float* out, int outStride, int batch = 1, long long inImageStride = 0, long long outImageStride = 0);
int bhip_launch_blur_fused(bhip_ctx* ctx, const float* kernel, int kw, const float* in, int inStride, int width, int height, float* out, int outStride, int batch,
						   long long inImageStride, long long outImageStride, bool* done);
int bhip_launch_pyr_layer_fused(bhip_ctx* ctx, const float* kernel, int kw, const float* in, long long inImageStride, int inStride, int width, int height,
								float* out, long long outImageStride, int outStride, int skip, int batch, bool* done);
int bhip_launch_conv_down(bhip_ctx* ctx, bool vertical, const float* kernel, int kw, const float* in, long long inImageStride, int inStride, int width,
						  int height, float* out, long long outImageStride, int outStride, int outWidth, int outHeight, int skip, int batch);
int bhip_launch_planar_average(bhip_ctx* ctx, const float* bands, long long bandStride, int numBands, long long n, float* out);
int bhip_launch_corner_intensity(bhip_ctx* ctx, int kind, int radius, float kappa, const float* dx, const float* dy, int dStride, int width, int height,
								 float* hXX, float* hXY, float* hYY, float* intensity, int iStride, int batch = 1, long long dImageStride = 0,
								 long long hImageStride = 0, long long iImageStride = 0);
int bhip_launch_conv2d(bhip_ctx* ctx, const float* kernel, int kw, int koff, const float* in, int inStride, int width, int height, float* out, int outStride);
int bhip_launch_mean(bhip_ctx* ctx, bool vertical, const float* in, float* out, int width, int height, int radius);
int bhip_launch_median(bhip_ctx* ctx, const float* in, int inStride, float* out, int outStride, int width, int height, int radius);
int bhip_launch_gradient(bhip_ctx* ctx, int kind, const float* in, int inStride, int width, int height, float* dx, float* dy, int outStride, int border,
						 int batch = 1, long long inImageStride = 0, long long outImageStride = 0);
int bhip_launch_grad_intensity(bhip_ctx* ctx, int kind, const float* dx, const float* dy, long long dImageStride, int dStride, float* out,
							   long long oImageStride, int oStride, int width, int height, int batch);
int bhip_launch_brief(bhip_ctx* ctx, const float* img, int stride, int width, int height, int radius, int numPoints, const int* samplePoints,
					  const int* compare, const double* xy, int n, int* out, bool u8 = false, int batch = 1, long long imageStride = 0,
					  const int* start = nullptr, int maxCount = 0, int xyStride = 2, long long xyImageStride = 0, bool patchOk = false);
// patchOk: every sample point of the definition lies within [-radius, radius]^2 (checked on the host where the table is at hand), so the
// LDS-patch kernel may be used; otherwise the gather kernel runs
// stand-alone strict block NMS over a batch (detect.hip): bitmap of accepted blocks + the pixel's position inside its block
int bhip_launch_nonmax_blocks(bhip_ctx* ctx, const float* img, long long imageStride, int stride, int w, int h, int batch, int radius, float threshold, int border,
							  unsigned int* bitmap, int bitmapWords, unsigned short* posInBlock, int nbx, int nby);
int bhip_launch_blocks_to_xy(bhip_ctx* ctx, const unsigned int* bitmap, const unsigned int* wordPrefix, int bitmapWords, const unsigned short* posInBlock,
							 int nbx, int nby, int batch, int radius, int border, int16_t* xy, int cap);

