// C ABI of libboofhip.so (include/boofhip.h): context, the SURF detect+describe object, association, stage-level entry points.
// Host orchestration only -- all arithmetic lives in the kernels.  There is deliberately no CPU fallback.
#include "common.h"
#include <cmath>
#include <cfloat>
#include <algorithm>
#include <memory>
#include <mutex>
#include <unordered_set>

// launchers defined in the other translation units
int bhip_launch_describe_ex(bhip_ctx* ctx, ImgView ii, const KeyPoint* kps, int cap, const int* imageStart, int batch, int singleImage, long long total,
							SurfTables t, const double* anglesIn, double* angles, double* desc, uint8_t* white, const int* perm = nullptr, const DescPlanar* planar = nullptr);
int bhip_launch_kp_spatial_order(bhip_ctx* ctx, const KeyPoint* kps, int cap, const int* start, int batch, int maxCount, int W, int H, int* hist, int* perm);
int bhip_assoc_phase1_l2(bhip_ctx* ctx, const double* src, int nsLocal, int srcBegin, const double* dst, int nd, int dof, double maxErr, int sqrtScore,
						 int* pairs, double* fit, void* colTop, DevBuf& work);
int bhip_assoc_phase1_ham(bhip_ctx* ctx, const int32_t* src, int nsLocal, int srcBegin, const int32_t* dst, int nd, int words, double maxErr, int* pairs,
						  double* fit, void* colTop, DevBuf& work);
int bhip_assoc_phase2(bhip_ctx* ctx, const void* colAll, int nranks, int nd, int nsLocal, int srcBegin, int* pairs, double* fit);
int bhip_assoc_coltop_size();
int bhip_assoc_hamming_batched(bhip_ctx* ctx, const int32_t* src, const int32_t* dst, int words, int count, const long long* srcOff, const int* ns,
							   const long long* dstOff, const int* nd, double maxErr, int backwards, int* pairs, double* fit, DevBuf& work);
int bhip_launch_integral_u8(bhip_ctx* ctx, const unsigned char* in, long long inImageStride, int inStride, int* out, long long outImageStride, int outStride,
							int width, int height, int batch);

int bhip_assoc_l2_mfma_batched(bhip_ctx* ctx, AssocMfmaWork& W, const double* dev_src, const double* dev_dst, int count, const long long* srcOff,
								 const int* ns, const long long* dstOff, const int* nd, double maxErr, int backwards, int* dev_pairs, double* dev_fit,
								 int* usedMfma);
void bhip_assoc_mfma_release(AssocMfmaWork& W);

bool bhip_fused_plan(int skip, int nlevels, const int* sizes, int radius, int* TX, int* TY, int* ldsBytes);
bool bhip_fused_is_fixed(int skip, int nlevels, const int* sizes, int radius);
int bhip_launch_detect_fused(bhip_ctx* ctx, ImgView ii, int batch, int skip, int nlevels, const int* sizes, int nmid, const DetectLevelParams* mids,
							 const int* midLevels, int radius, float threshold, unsigned int* bitmap, int bitmapWords, KeyPoint* cand, int* candCount,
							 int cap, const FusedExport* exp, bool intTaps = false);

// per-context scratch that the stateless entry points reuse
struct CtxScratch {
	DevBuf a, b, c, d, e, work, nmsBitmap, nmsPrefix, nmsPos, ipTmp, ipKernel;
	AssocMfmaWork mfma;
	int assocExactOnly = -1;  // BHIP_ASSOC_EXACT=1 forces the exact VALU association kernels (parity cross-check)
};
static CtxScratch* scratchOf(bhip_ctx* ctx);

struct bhip_ctx_full : bhip_ctx {
	CtxScratch scratch;
};
static CtxScratch* scratchOf(bhip_ctx* ctx) { return &static_cast<bhip_ctx_full*>(ctx)->scratch; }

// Handle registry: which bhip_ctx / bhip_surf pointers are live.  The destroy calls may arrive in any order (a garbage-collected host
// language finalises objects in no particular order; a caller may close the context first) and more than once: destroying a context
// releases the device side of every detect+describe object created on it and leaves those objects as inert shells (every call on them
// returns BHIP_ERR_INVALID, bhip_surf_destroy only frees the shell); a pointer that is not in the registry is refused instead of
// dereferenced.  Once the process has started to exit (atexit) the destroy calls touch neither the HIP runtime nor the handles -- the
// runtime's own teardown may already have run.  The registry is a leaked singleton so it outlives every static destructor.
struct HandleRegistry {
	std::mutex m;
	std::unordered_set<bhip_ctx*> ctxs;
	std::unordered_set<bhip_surf*> surfs;
	bool exiting = false;
};
static HandleRegistry& registry() {
	static HandleRegistry* r = [] {
		HandleRegistry* p = new HandleRegistry();
		atexit([] { registry().exiting = true; });   // registered after the HIP runtime loaded, so it runs before the runtime's teardown
		return p;
	}();
	return *r;
}
static void surfOrphanChildren(bhip_ctx* ctx);   // registry lock held: releases the device side of every live bhip_surf created on ctx

extern "C" {

const char* bhip_version(void) { return "boofhip 0.1 (gfx950)"; }

void bhip_fh_cfg_default(bhip_fh_cfg* c) {
	c->detectThreshold = 1; c->extractRadius = 2; c->maxFeaturesPerScale = -1; c->initialSampleSize = 1; c->initialSize = 9;
	c->numberScalesPerOctave = 4; c->numberOfOctaves = 4; c->scaleStepSize = 6;
}
void bhip_surf_cfg_default(bhip_surf_cfg* c) {
	c->widthLargeGrid = 4; c->widthSubRegion = 5; c->widthSample = 3; c->weightSigma = 4.5; c->overLap = 2; c->sigmaLargeGrid = 2.5;
	c->sigmaSubRegion = 2.5;
}
void bhip_ori_cfg_default(bhip_ori_cfg* c, int stable) {
	c->objectRadiusToScale = 1.0 / 2.0;
	c->weightSigma = -1;
	c->sampleWidth = 6;
	if (stable) { c->samplePeriod = 0.65; c->windowSize = M_PI / 3.0; c->radius = 8; }
	else { c->samplePeriod = 1; c->windowSize = 0; c->radius = 6; }
}

static int ctxCreate(int device, void* stream, bool useGiven, bhip_ctx** out) {
	if (!out) return BHIP_ERR_INVALID;
	*out = nullptr;
	int count = 0;
	if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return BHIP_ERR_HIP;  // no GPU: fail loudly, never fall back
	if (device < 0 || device >= count) return BHIP_ERR_INVALID;
	if (hipSetDevice(device) != hipSuccess) return BHIP_ERR_HIP;
	bhip_ctx_full* ctx = new (std::nothrow) bhip_ctx_full();
	if (!ctx) return BHIP_ERR_NOMEM;
	ctx->device = device;
	if (useGiven) {
		ctx->stream = (hipStream_t)stream;
		ctx->ownStream = false;
	} else {
		if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return BHIP_ERR_HIP; }
		ctx->ownStream = true;
	}
	if (hipHostMalloc((void**)&ctx->hostScratch, 1 << 20, hipHostMallocDefault) != hipSuccess) {
		if (ctx->ownStream) (void)hipStreamDestroy(ctx->stream);
		delete ctx;
		return BHIP_ERR_HIP;
	}
	{
		HandleRegistry& R = registry();
		std::lock_guard<std::mutex> lock(R.m);
		R.ctxs.insert(ctx);
	}
	*out = ctx;
	return BHIP_OK;
}
int bhip_ctx_create(int device, bhip_ctx** out) { return ctxCreate(device, nullptr, false, out); }
int bhip_ctx_create_on_stream(int device, void* hip_stream, bhip_ctx** out) { return ctxCreate(device, hip_stream, true, out); }
int bhip_ctx_destroy(bhip_ctx* c) {
	if (!c) return BHIP_OK;
	HandleRegistry& R = registry();
	std::lock_guard<std::mutex> lock(R.m);
	if (R.exiting) return BHIP_OK;                        // process teardown: the runtime reclaims everything
	if (!R.ctxs.count(c)) return BHIP_ERR_INVALID;        // not a live context (destroyed twice, or never created)
	R.ctxs.erase(c);
	bhip_ctx_full* ctx = static_cast<bhip_ctx_full*>(c);
	(void)hipSetDevice(ctx->device);
	(void)hipStreamSynchronize(ctx->stream);
	// detect+describe objects still alive on this context lose their device side now and become inert shells
	surfOrphanChildren(c);
	CtxScratch& s = ctx->scratch;
	s.a.release(); s.b.release(); s.c.release(); s.d.release(); s.e.release(); s.work.release();
	s.nmsBitmap.release(); s.nmsPrefix.release(); s.nmsPos.release(); s.ipTmp.release(); s.ipKernel.release();
	bhip_assoc_mfma_release(s.mfma);
	bhip_profile_release(ctx);
	if (ctx->hostScratch) (void)hipHostFree(ctx->hostScratch);
	if (ctx->ownStream) (void)hipStreamDestroy(ctx->stream);
	delete ctx;
	return BHIP_OK;
}
int bhip_ctx_synchronize(bhip_ctx* ctx) {
	if (!ctx) return BHIP_ERR_INVALID;
	BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return BHIP_OK;
}
const char* bhip_last_error(bhip_ctx* ctx) { return ctx ? ctx->error.c_str() : "null context"; }
// Page-locked host memory for the arrays that cross the boundary (results of bhip_surf_fetch, descriptor lists handed to
// bhip_assoc_l2_f64, frames): copies to and from such memory are DMA transfers at PCIe speed, copies from pageable memory go through
// the runtime's staging buffer (1.1 MB of descriptors: ~0.15 ms instead of ~0.03 ms).  The block belongs to the process, not to the
// context: it stays valid after bhip_ctx_destroy and is released by bhip_host_free (after exit has begun: by the runtime).
int bhip_host_alloc(bhip_ctx* ctx, long long bytes, uint8_t** host_mem) {
	if (!ctx || !host_mem) return BHIP_ERR_INVALID;
	*host_mem = nullptr;
	if (bytes <= 0) return bhip_fail(ctx, BHIP_ERR_INVALID, "bhip_host_alloc: size must be positive");
	BHIP_HIP(ctx, hipSetDevice(ctx->device));
	void* p = nullptr;
	const hipError_t e = hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault);
	if (e != hipSuccess) { (void)hipGetLastError(); return bhip_fail(ctx, BHIP_ERR_NOMEM, "bhip_host_alloc: out of page-locked memory"); }
	*host_mem = (uint8_t*)p;
	return BHIP_OK;
}
int bhip_host_free(void* host_mem) {
	if (!host_mem) return BHIP_OK;
	{
		HandleRegistry& R = registry();
		std::lock_guard<std::mutex> lock(R.m);
		if (R.exiting) return BHIP_OK;   // process teardown: the runtime reclaims it
	}
	return hipHostFree(host_mem) == hipSuccess ? BHIP_OK : BHIP_ERR_HIP;
}

}  // extern "C"

#define CHECK_CTX(ctx)                                   \
	do {                                                 \
		if (!(ctx)) return BHIP_ERR_INVALID;             \
		BHIP_HIP((ctx), hipSetDevice((ctx)->device));    \
	} while (0)

// ---------------------------------------------------------------------------------------------------------------
// Fast-Hessian detector: octave schedule + buffers (FastHessianFeatureDetector.detect :156-188, detectOctave :198-221)
// ---------------------------------------------------------------------------------------------------------------
struct FhLevelPlan {
	DetectLevelParams p;
	int level;  // index of the mid level inside its octave
};
struct FhOctavePlan {
	int skip, w, h, nlevels;
	int sizes[BHIP_MAX_LEVELS];
	std::vector<FhLevelPlan> mids;
	// execution plan (FhDetector::planExecution)
	bool fused = false, fixed = false;
	int shareFrom[BHIP_MAX_LEVELS];   // level of the previous octave with the same kernel size, or -1
	bool onDemand[BHIP_MAX_LEVELS];   // stand-alone octave: outer level left to k_nms_scalespace (evaluated around the NMS maxima only)
	int exportSlot[BHIP_MAX_LEVELS];  // fused producer: slot of this level in the export buffer, or -1
	int nexport = 0;
	size_t intenOff = 0, expOff = 0;  // floats, per-image strides below
	long long intenImageStride = 0, expImageStride = 0;
};

struct FhDetector {
	bhip_fh_cfg cfg;
	int W = 0, H = 0, batch = 0, cap = 0;
	bool intTaps = false;   // the integral image holds int32 (GrayS32, from a GrayU8 frame) instead of float
	std::vector<FhOctavePlan> plan;
	int bitmapWords = 0;
	DevBuf inten, expBuf, bitmap, prefix, cand, sorted, count, selKey, selIdx, selLevels;
	bool nBest() const { return cfg.maxFeaturesPerScale > 0; }   // SelectNBestFeatures between the NMS and the scale-space test
	std::vector<int> counts;   // per image, host
	long long total = 0;

	static bool unfusedOnly() { return bhip_env_flag("BHIP_DETECT_UNFUSED"); }   // parity cross-check of the two detector paths
	int makePlan(bhip_ctx* ctx, int width, int height) {
		plan.clear();
		if (cfg.numberScalesPerOctave > BHIP_MAX_LEVELS || cfg.numberScalesPerOctave < 1) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "numberScalesPerOctave out of range");
		if (cfg.extractRadius < 1) return bhip_fail(ctx, BHIP_ERR_INVALID, "Search radius must be >= 1");
		if (cfg.initialSampleSize < 1 || cfg.initialSize < 3) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad sample size / initial size");
		if (width >= 32768 || height >= 32768) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "image too large for Point2D_I16");
		int skip = cfg.initialSampleSize, sizeStep = cfg.scaleStepSize, octaveSize = cfg.initialSize;
		unsigned int bits = 0;
		const int step = cfg.extractRadius + 1;
		for (int octave = 0; octave < cfg.numberOfOctaves; octave++) {
			FhOctavePlan o;
			o.nlevels = cfg.numberScalesPerOctave;
			for (int i = 0; i < o.nlevels; i++) o.sizes[i] = octaveSize + i * sizeStep;
			const int maxSize = o.sizes[o.nlevels - 1];
			if (maxSize > width || maxSize > height) break;
			o.skip = skip;
			o.w = width / skip;
			o.h = height / skip;
			for (int i = 2; i < o.nlevels; i++) {
				FhLevelPlan m;
				m.level = i - 1;
				DetectLevelParams& p = m.p;
				p.skip = skip; p.w = o.w; p.h = o.h;
				p.sizeLower = o.sizes[m.level - 1]; p.sizeMid = o.sizes[m.level]; p.sizeUpper = o.sizes[m.level + 1];
				p.border = p.sizeMid / (2 * skip);
				const int rw = p.w - 2 * p.border, rh = p.h - 2 * p.border;
				p.nbx = rw > 0 ? (rw + step - 1) / step : 0;
				p.nby = rh > 0 ? (rh + step - 1) / step : 0;
				p.bitBase = bits;
				bits += (unsigned)p.nbx * (unsigned)p.nby;
				o.mids.push_back(m);
			}
			plan.push_back(o);
			skip += skip;
			octaveSize += sizeStep;
			sizeStep += sizeStep;
		}
		bitmapWords = (int)((bits + 31) / 32) + 1;
		return BHIP_OK;
	}

	static bool noShare() { return bhip_env_flag("BHIP_DETECT_NOSHARE"); }   // parity cross-check of the shared-level plan
	static bool denseOuter() { return bhip_env_flag("BHIP_DETECT_DENSE"); }  // parity cross-check: compute the outer levels of every octave densely
	// Which octaves run fused, and which levels are copied from the octave below instead of being recomputed.  A box-filter response
	// depends on (pixel, kernel size) only, and the default schedule repeats sizes: 15,27 | 27,51 | 51,99 are levels 1,3 of one octave and
	// levels 0,1 of the next, on a lattice twice as coarse.  Sharing is enabled where the unrolled inner form and the clamped border
	// form of the reference cover the same boxes (size = 3*blockSmall, odd), between a producer that keeps or exports its intensity and a
	// stand-alone consumer.  The two forms still round Dyy differently, and which pixels are "inner" depends on the step, so the
	// consumer (k_hessian) copies a pixel only when both octaves evaluate it with the same form and computes the rest itself.
	void planExecution() {
		for (auto& o : plan) {
			int ftx, fty, flds;
			// N-best selection needs every level's NMS list and intensity images in memory: stand-alone kernels
			o.fused = !unfusedOnly() && !nBest() && !o.mids.empty() && bhip_fused_plan(o.skip, o.nlevels, o.sizes, cfg.extractRadius, &ftx, &fty, &flds);
			o.fixed = o.fused && bhip_fused_is_fixed(o.skip, o.nlevels, o.sizes, cfg.extractRadius);
			if (intTaps && !o.fixed) o.fused = false;   // integer taps: compile-time-geometry fused kernel or the stand-alone kernels
			o.nexport = 0;
			for (int i = 0; i < BHIP_MAX_LEVELS; i++) { o.shareFrom[i] = -1; o.exportSlot[i] = -1; o.onDemand[i] = false; }
		}
		// The first and last level of an octave are only read around the NMS maxima of their neighbour level: a stand-alone octave leaves
		// them to k_nms_scalespace unless they can be copied from the octave below.  (N-best selection reads whole levels: dense.)
		const bool sparseOuter = !denseOuter() && !nBest();
		for (size_t k = 0; k < plan.size(); k++) {
			FhOctavePlan& c = plan[k];
			if (k > 0 && !noShare()) shareLevels(c, plan[k - 1]);
			if (!c.fused && sparseOuter && c.nlevels >= 3)
				for (int i : {0, c.nlevels - 1})
					if (c.shareFrom[i] < 0) c.onDemand[i] = true;
		}
	}
	void shareLevels(FhOctavePlan& c, FhOctavePlan& p) {
		{
			if (c.fused || c.skip != 2 * p.skip) return;
			if (p.fused && !p.fixed) return;
			for (int i = 0; i < c.nlevels; i++) {
				const int size = c.sizes[i];
				if (size % 3 != 0 || size % 2 != 1) continue;
				for (int j = 0; j < p.nlevels; j++) {
					if (p.sizes[j] != size) continue;
					if (!p.fused && p.onDemand[j]) break;   // the producer does not hold this level
					if (p.fused) {
						if (p.exportSlot[j] < 0) {
							if (p.nexport >= 2) break;
							p.exportSlot[j] = p.nexport++;
						}
					}
					c.shareFrom[i] = j;
					break;
				}
			}
		}
	}

	bool plannedInt = false;
	int prepare(bhip_ctx* ctx, int width, int height, int batch_) {
		if (width != W || height != H || plannedInt != intTaps) { BHIP_TRY(makePlan(ctx, width, height)); planExecution(); plannedInt = intTaps; }
		W = width; H = height; batch = batch_;
		if (cap == 0) cap = 8192;
		return allocate(ctx);
	}
	int allocate(bhip_ctx* ctx) {
		size_t intenFloats = 0;
		for (size_t k = 0; k < plan.size(); k++) {
			FhOctavePlan& o = plan[k];
			if (!o.fused) {
				o.intenImageStride = (long long)o.nlevels * o.w * o.h;
				o.intenOff = intenFloats;
				intenFloats += (size_t)o.intenImageStride * batch;
			}
		}
		BHIP_TRY(inten.reserve(ctx, intenFloats * 4 + 16));
		BHIP_TRY(bitmap.reserve(ctx, (size_t)bitmapWords * 4 * batch));
		BHIP_TRY(prefix.reserve(ctx, (size_t)bitmapWords * 4 * batch));
		BHIP_TRY(cand.reserve(ctx, (size_t)cap * sizeof(KeyPoint) * batch));
		BHIP_TRY(sorted.reserve(ctx, (size_t)cap * sizeof(KeyPoint) * batch));
		BHIP_TRY(count.reserve(ctx, (size_t)batch * 4 * 2));
		if (nBest()) {
			size_t nlv = 0;
			for (auto& o : plan) nlv += o.mids.size();
			BHIP_TRY(selKey.reserve(ctx, (size_t)cap * 4 * batch));
			BHIP_TRY(selIdx.reserve(ctx, (size_t)cap * 4 * batch));
			BHIP_TRY(selLevels.reserve(ctx, std::max<size_t>(nlv, 1) * 4 * 2 * batch));
		}
		return BHIP_OK;
	}

	// ii: dense integral images.  Leaves the ordered key points in `sorted` ([image][cap]) and their counts in `counts`.
	int run(bhip_ctx* ctx, ImgView ii) {
		for (int attempt = 0; attempt < 8; attempt++) {
			BHIP_HIP(ctx, hipMemsetAsync(bitmap.p, 0, (size_t)bitmapWords * 4 * batch, ctx->stream));
			BHIP_HIP(ctx, hipMemsetAsync(count.p, 0, (size_t)batch * 4 * 2, ctx->stream));
			for (size_t k = 0; k < plan.size(); k++) {
				FhOctavePlan& o = plan[k];
				if (o.fused) {
					// LDS-tiled fused octave: intensity never leaves the CU (except the levels the next octave shares)
					DetectLevelParams mp[BHIP_MAX_LEVELS];
					int ml[BHIP_MAX_LEVELS];
					for (size_t q = 0; q < o.mids.size(); q++) { mp[q] = o.mids[q].p; ml[q] = o.mids[q].level; }
					FusedExport ex{0, {0, 0}, nullptr, 0, 0, 0, {0, 0}};
					if (o.nexport > 0 && k + 1 < plan.size()) {
						// the shared levels go straight into the consuming octave's level planes: its k_hessian then only recomputes the
						// pixels the two octaves evaluate with different forms
						const FhOctavePlan& c = plan[k + 1];
						ex.n = o.nexport;
						for (int j = 0; j < o.nlevels; j++) if (o.exportSlot[j] >= 0) ex.level[o.exportSlot[j]] = j;
						ex.out = inten.as<float>() + c.intenOff; ex.w = c.w; ex.h = c.h; ex.imageStride = c.intenImageStride;
						for (int i = 0; i < c.nlevels; i++) {
							const int j = c.shareFrom[i];
							if (j >= 0 && o.exportSlot[j] >= 0) ex.slotOffset[o.exportSlot[j]] = (long long)i * c.w * c.h;
						}
					}
					BHIP_TRY(bhip_launch_detect_fused(ctx, ii, batch, o.skip, o.nlevels, o.sizes, (int)o.mids.size(), mp, ml, cfg.extractRadius,
													  cfg.detectThreshold, bitmap.as<unsigned int>(), bitmapWords, cand.as<KeyPoint>(), count.as<int>(), cap,
													  ex.n > 0 ? &ex : nullptr, intTaps));
					continue;
				}
				const long long levelStride = (long long)o.w * o.h;
				const long long imageStride = o.intenImageStride;
				float* base = inten.as<float>() + o.intenOff;
				HessLevelSource from[BHIP_MAX_LEVELS];
				for (int i = 0; i < o.nlevels; i++) {
					from[i] = HessLevelSource{nullptr, 0, 0, 1, 0};
					const int j = o.shareFrom[i];
					if (j < 0 || k == 0) continue;
					const FhOctavePlan& p = plan[k - 1];
					if (p.fused) from[i] = HessLevelSource{base + (size_t)i * levelStride, imageStride, o.w, 1, 1};   // written in place by the fused octave
					else from[i] = HessLevelSource{inten.as<float>() + p.intenOff + (size_t)j * p.w * p.h, p.intenImageStride, p.w, 2, 0};
				}
				unsigned int skipMask = 0;
				for (int i = 0; i < o.nlevels; i++)
					if (o.onDemand[i]) skipMask |= 1u << i;
				BHIP_TRY(bhip_launch_hessian(ctx, ii, batch, o.skip, o.nlevels, o.sizes, base, levelStride, imageStride, o.w, from, intTaps, skipMask));
				for (auto& m : o.mids) {
					const float* lower = o.onDemand[m.level - 1] ? nullptr : base + (m.level - 1) * levelStride;
					const float* upper = o.onDemand[m.level + 1] ? nullptr : base + (m.level + 1) * levelStride;
					BHIP_TRY(bhip_launch_nms_scalespace(ctx, lower, base + m.level * levelStride, upper, imageStride, o.w, batch, m.p, cfg.extractRadius,
														cfg.detectThreshold, bitmap.as<unsigned int>(), bitmapWords, cand.as<KeyPoint>(), count.as<int>(), cap,
														nBest(), &ii, intTaps));
				}
			}
			BHIP_TRY(bhip_launch_word_prefix(ctx, bitmap.as<unsigned int>(), bitmapWords, batch, prefix.as<unsigned int>(), count.as<int>() + batch));
			counts.resize(batch);
			if ((size_t)batch * 4 <= (1u << 20)) {
				BHIP_HIP(ctx, hipMemcpyAsync(ctx->hostScratch, count.p, (size_t)batch * 4, hipMemcpyDeviceToHost, ctx->stream));
				BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
				memcpy(counts.data(), ctx->hostScratch, (size_t)batch * 4);
			} else {
				BHIP_HIP(ctx, hipMemcpyAsync(counts.data(), count.p, (size_t)batch * 4, hipMemcpyDeviceToHost, ctx->stream));
				BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
			}
			int maxCount = 0;
			total = 0;
			for (int c : counts) { maxCount = std::max(maxCount, c); total += c; }
			if (maxCount <= cap) {
				BHIP_TRY(bhip_launch_rank_scatter(ctx, bitmap.as<unsigned int>(), bitmapWords, prefix.as<unsigned int>(), cand.as<KeyPoint>(), count.as<int>(),
												  cap, batch, sorted.as<KeyPoint>()));
				if (nBest()) BHIP_TRY(selectNBest(ctx));
				return BHIP_OK;
			}
			// candidate list overflowed: grow and run the detector again
			cap = maxCount + maxCount / 4 + 64;
			BHIP_TRY(allocate(ctx));
		}
		return bhip_fail(ctx, BHIP_ERR_CAPACITY, "key point list kept overflowing");
	}
	// maxFeaturesPerScale > 0: `sorted` holds every level's NMS maxima (ranked, with intensities).  Per level: select, scale-space test,
	// sub-pixel fit (k_select_nbest, results into `cand`), then the levels are packed back into `sorted` and the counts re-read.
	int selectNBest(bhip_ctx* ctx) {
		int nlv = 0;
		for (auto& o : plan) nlv += (int)o.mids.size();
		int* levelStart = selLevels.as<int>();
		int* levelCount = levelStart + (size_t)std::max(nlv, 1) * batch;
		int li = 0;
		for (auto& o : plan) {
			const long long levelStride = (long long)o.w * o.h;
			const float* base = inten.as<float>() + o.intenOff;
			for (auto& m : o.mids) {
				BHIP_TRY(bhip_launch_select_nbest(ctx, base + (m.level - 1) * levelStride, base + m.level * levelStride, base + (m.level + 1) * levelStride,
												  o.intenImageStride, o.w, batch, m.p, cfg.extractRadius, cfg.maxFeaturesPerScale, bitmap.as<unsigned int>(),
												  prefix.as<unsigned int>(), bitmapWords, sorted.as<KeyPoint>(), cap, selKey.as<float>(), selIdx.as<int>(),
												  cand.as<KeyPoint>(), levelStart, levelCount, li, nlv));
				li++;
			}
		}
		BHIP_TRY(bhip_launch_compact_levels(ctx, cand.as<KeyPoint>(), cap, levelStart, levelCount, nlv, batch, sorted.as<KeyPoint>(), count.as<int>()));
		BHIP_HIP(ctx, hipMemcpyAsync(counts.data(), count.p, (size_t)batch * 4, hipMemcpyDeviceToHost, ctx->stream));
		BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
		total = 0;
		for (int c : counts) total += c;
		return BHIP_OK;
	}
	void release() {
		inten.release(); expBuf.release(); bitmap.release(); prefix.release(); cand.release(); sorted.release(); count.release();
		selKey.release(); selIdx.release(); selLevels.release();
	}
};

// ---------------------------------------------------------------------------------------------------------------
// SURF detect + describe object
// ---------------------------------------------------------------------------------------------------------------
struct bhip_surf {
	bhip_ctx* ctx = nullptr;
	int stable = 1;
	bhip_surf_cfg sd;
	bhip_ori_cfg ori;
	FhDetector det;
	SurfTables tables;
	DevBuf tabBuf, inBuf, iiBuf, startBuf, angBuf, descBuf, whiteBuf, xysBuf, tmpKp, tmpAng, tmpDesc, tmpWhite, permBuf;
	std::vector<int> starts;  // batch+1
	int* startsPinned = nullptr;          // page-locked staging copy of `starts` for its upload
	size_t startsPinnedBytes = 0;
	int W = 0, H = 0, batch = 0;
	bool haveResult = false;
	ImgView iiView;
	int planarBands = 0;       // > 0: the last detect was colour SURF on that many bands (descriptor = planarBands * dof values)
	DescPlanar planar{};
	bool descOptions = false;  // the last detect needs `planar` passed to the describe kernel (colour bands and / or integer taps)
	int dofOut() const { return tables.dof * (planarBands > 0 ? planarBands : 1); }
	// host-frame batches are processed in chunks so that the upload of chunk k+1 (copy stream) runs under the kernels of chunk k: the
	// chunks go through `worker` (same configuration, chunk-sized work buffers), whose results are appended to this object's arrays
	bhip_surf* worker = nullptr;
	hipStream_t copyStream = nullptr;
	float* extII = nullptr;    // worker only: where the integral images of the current chunk go (a slice of the owner's iiBuf)
	// describe = BRIEF (DetectDescribeFusion(fastHessian, null, brief), bhip_surf_create_brief): no orientation / SURF stage; every detected
	// point gets its TupleDesc_B words from the input frame itself
	bool brief = false;
	int briefRadius = 0, briefPoints = 0, briefWords = 0;
	DevBuf briefTab, wordsBuf;   // [samplePoints | compare] on the device; words of the whole batch, compact [total][briefWords]
	bool briefPatch = false;            // the definition fits the LDS-patch kernel
	const int* briefBorrow = nullptr;   // chunk worker: the owner's table
	const int* briefSample() const { return briefBorrow ? briefBorrow : briefTab.as<int>(); }
	const int* briefCompare() const { return briefSample() + briefCompareOff; }
	size_t briefCompareOff = 0;
};

static int buildTables(bhip_surf* s) {
	bhip_ctx* ctx = s->ctx;
	SurfTables& t = s->tables;
	memset(&t, 0, sizeof(t));
	const bhip_surf_cfg& c = s->sd;
	const bhip_ori_cfg& o = s->ori;
	if (c.widthLargeGrid < 1 || c.widthSubRegion < 1 || c.widthSample < 1 || c.overLap < 0) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad SURF config");
	if (o.radius < 1 || o.radius > 10) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "orientation radius out of range (1..10 on the GPU)");
	std::vector<double> all;
	auto push = [&](const std::vector<double>& v) { size_t off = all.size(); all.insert(all.end(), v.begin(), v.end()); return off; };
	t.oriStable = s->stable;
	t.oriRadius = o.radius;
	t.oriWidth = 2 * o.radius + 1;
	t.oriKernelWidth = o.sampleWidth;
	t.oriHasWeights = o.weightSigma != 0 ? 1 : 0;
	t.oriPeriod = o.samplePeriod;
	t.oriWindow = o.windowSize;
	t.oriRadiusToScale = o.objectRadiusToScale;
	size_t offOri = 0, offSub = 0, offGrid = 0, offFast = 0;
	// OrientationIntegralBase :85-86 weights = FactoryKernelGaussian.gaussian(2,true,64,weightSigma,sampleRadius)
	if (t.oriHasWeights) offOri = push(bhip_gaussian2d_f64(o.weightSigma, o.radius));
	t.stable = s->stable;
	t.widthLargeGrid = c.widthLargeGrid; t.widthSubRegion = c.widthSubRegion; t.widthSample = c.widthSample; t.overLap = c.overLap;
	t.dof = c.widthLargeGrid * c.widthLargeGrid * 4;
	const int regionSize = c.widthLargeGrid * c.widthSubRegion;
	if (s->stable) {
		// DescribePointSurfMod :76-106
		std::vector<double> wg = bhip_gaussian_width(c.sigmaLargeGrid, c.widthLargeGrid);
		std::vector<double> ws = bhip_gaussian_width(c.sigmaSubRegion, c.widthSubRegion + 2 * c.overLap);
		const int gw = c.widthLargeGrid, sw = c.widthSubRegion + 2 * c.overLap;
		double div = wg[(size_t)(gw / 2) * gw + gw / 2];
		for (double& v : wg) v /= div;
		div = ws[(size_t)(sw / 2) * sw + sw / 2];
		for (double& v : ws) v /= div;
		offGrid = push(wg);
		offSub = push(ws);
		t.radiusDescriptor = regionSize / 2 + c.overLap;
	} else {
		// DescribePointSurf :110-141
		const int radius = regionSize / 2;
		std::vector<double> w = bhip_gaussian_width(c.weightSigma, radius * 2);
		if ((int)w.size() != regionSize * regionSize) return bhip_fail(ctx, BHIP_ERR_INVALID, "Weighting kernel has an unexpected size");
		const double div = w[(size_t)radius * (radius * 2) + radius];
		for (double& v : w) v /= div;
		offFast = push(w);
		t.radiusDescriptor = regionSize / 2;
	}
	BHIP_TRY(s->tabBuf.reserve(ctx, all.size() * 8 + 8));
	BHIP_HIP(ctx, hipMemcpyAsync(s->tabBuf.p, all.data(), all.size() * 8, hipMemcpyHostToDevice, ctx->stream));
	BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
	const double* base = s->tabBuf.as<double>();
	t.oriWeights = t.oriHasWeights ? base + offOri : nullptr;
	t.weightSub = s->stable ? base + offSub : nullptr;
	t.weightGrid = s->stable ? base + offGrid : nullptr;
	t.weightFast = s->stable ? nullptr : base + offFast;
	return BHIP_OK;
}

// planarBands > 0: `in` holds [1 + planarBands] images -- the band average first, then the bands (colour SURF, one frame)
// u8: `in.data` points at dense 8-bit frames ([batch][H][W] bytes); the integral images are then GrayS32 and every stage runs on integer taps
// deviceInput: the frames are the caller's device-resident batch (bhip_surf_detect_dev_f32) -- the call may return while the describe kernels
// are still queued on the context's stream (its documented contract); host-frame callers get their frames back only after a final synchronize
static int surfRun(bhip_surf* s, ImgView in, int batch, int planarBands = 0, bool u8 = false, bool deviceInput = false) {
	bhip_ctx* ctx = s->ctx;
	const int W = in.width, H = in.height;
	s->haveResult = false;
	s->det.intTaps = u8;
	BHIP_TRY(s->det.prepare(ctx, W, H, batch));
	const int nImages = planarBands > 0 ? 1 + planarBands : batch;
	float* iiBase = s->extII;
	if (!iiBase) {
		BHIP_TRY(s->iiBuf.reserve(ctx, (size_t)W * H * 4 * nImages));
		iiBase = s->iiBuf.as<float>();
	}
	s->W = W; s->H = H; s->batch = batch;
	ImgViewW iiW{iiBase, (long long)W * H, W, W, H};
	if (u8) BHIP_TRY(bhip_launch_integral_u8(ctx, (const unsigned char*)in.data, (long long)W * H, W, (int*)iiBase, (long long)W * H, W, W, H, nImages));
	else BHIP_TRY(bhip_launch_integral(ctx, in, iiW, nImages));
	ImgView ii{iiBase, (long long)W * H, W, W, H};
	s->iiView = ii;
	s->planarBands = planarBands;
	s->planar = DescPlanar{iiBase + (long long)W * H, (long long)W * H * (1 + planarBands), (long long)W * H, planarBands,
						   planarBands > 0 ? 1.0 : 2.0, u8};
	s->descOptions = planarBands > 0 || u8;
	BHIP_TRY(s->det.run(ctx, ii));
	// exclusive prefix of counts -> start of every image in the compact result arrays
	s->starts.assign(batch + 1, 0);
	for (int i = 0; i < batch; i++) s->starts[i + 1] = s->starts[i] + s->det.counts[i];
	const long long total = s->det.total;
	if (total > 0x7fffffffLL) return bhip_fail(ctx, BHIP_ERR_CAPACITY, "more than 2^31 key points in one batch");
	BHIP_TRY(s->startBuf.reserve(ctx, (size_t)(batch + 1) * 4));
	{
		// through a page-locked staging copy owned by the object: the transfer then reads it in stream order, and nothing rewrites it before the
		// next detect has synchronized on its own count read-back
		const size_t need = (size_t)(batch + 1) * 4;
		if (need > s->startsPinnedBytes) {
			if (s->startsPinned) { BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream)); (void)hipHostFree(s->startsPinned); s->startsPinned = nullptr; s->startsPinnedBytes = 0; }
			if (hipHostMalloc((void**)&s->startsPinned, need * 2, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return bhip_fail(ctx, BHIP_ERR_NOMEM, "page-locked staging buffer"); }
			s->startsPinnedBytes = need * 2;
		}
		memcpy(s->startsPinned, s->starts.data(), need);
		BHIP_HIP(ctx, hipMemcpyAsync(s->startBuf.p, s->startsPinned, need, hipMemcpyHostToDevice, ctx->stream));
	}
	if (s->brief) {
		// DetectDescribeFusion.detect (F:abst/feature/detdesc/DetectDescribeFusion.java:95-127) with orientation == null and
		// describe = WrapDescribeBrief (process always returns true, so every detected point is kept, in detector order):
		// yaw = detector.getOrientation(i) = 0 (WrapFHtoInterestPoint.java:77-79); DescribePointBrief.process samples the frame handed to
		// setImage (:71-75,86-88 -- the blurred copy it makes is never read)
		if (planarBands > 0) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "BRIEF runs on single-band frames");
		BHIP_TRY(s->angBuf.reserve(ctx, (size_t)std::max<long long>(total, 1) * 8));
		BHIP_TRY(s->whiteBuf.reserve(ctx, (size_t)std::max<long long>(total, 1)));
		BHIP_TRY(s->wordsBuf.reserve(ctx, (size_t)std::max<long long>(total, 1) * 4 * s->briefWords));
		if (total > 0) {
			BHIP_HIP(ctx, hipMemsetAsync(s->angBuf.p, 0, (size_t)total * 8, ctx->stream));
			BHIP_HIP(ctx, hipMemsetAsync(s->whiteBuf.p, 0, (size_t)total, ctx->stream));
			int maxCount = 0;
			for (int c : s->det.counts) maxCount = std::max(maxCount, c);
			BHIP_TRY(bhip_launch_brief(ctx, in.data, u8 ? W : in.stride, W, H, s->briefRadius, s->briefPoints, s->briefSample(), s->briefCompare(),
									   (const double*)s->det.sorted.p, (int)total, s->wordsBuf.as<int>(), u8, batch, u8 ? (long long)W * H : in.imageStride,
									   s->startBuf.as<int>(), maxCount, (int)(sizeof(KeyPoint) / 8), (long long)s->det.cap * (long long)(sizeof(KeyPoint) / 8), s->briefPatch));
		}
		BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
		s->haveResult = true;
		return BHIP_OK;
	}
	const int dof = s->dofOut();
	BHIP_TRY(s->angBuf.reserve(ctx, (size_t)std::max<long long>(total, 1) * 8));
	BHIP_TRY(s->descBuf.reserve(ctx, (size_t)std::max<long long>(total, 1) * 8 * dof));
	BHIP_TRY(s->whiteBuf.reserve(ctx, (size_t)std::max<long long>(total, 1)));
	// processing order of the describe stage: key points of one image grouped by coarse tile (L2 locality of the gathers)
	const int* perm = nullptr;
	{
#ifdef BHIP_EXPERIMENTS
		const bool noOrder = bhip_env_flag("BHIP_DESCRIBE_NOORDER");
#else
		const bool noOrder = false;
#endif
		int maxCount = 0;
		for (int c : s->det.counts) maxCount = std::max(maxCount, c);
		if (!noOrder && total > 0) {
			BHIP_TRY(s->permBuf.reserve(ctx, (size_t)total * 4 + (size_t)batch * 64 * 4));
			int* hist = s->permBuf.as<int>() + total;
			BHIP_TRY(bhip_launch_kp_spatial_order(ctx, s->det.sorted.as<KeyPoint>(), s->det.cap, s->startBuf.as<int>(), batch, maxCount, W, H, hist,
												  s->permBuf.as<int>()));
			perm = s->permBuf.as<int>();
		}
	}
	BHIP_TRY(bhip_launch_describe_ex(ctx, ii, s->det.sorted.as<KeyPoint>(), s->det.cap, s->startBuf.as<int>(), batch, 0, total, s->tables, nullptr,
									 s->angBuf.as<double>(), s->descBuf.as<double>(), s->whiteBuf.as<uint8_t>(), perm, s->descOptions ? &s->planar : nullptr));
	// host-frame callers may reuse their frames (and the upload buffer) as soon as the call returns; a device-resident batch was consumed before
	// the detector's count read-back, so its call returns with the describe kernels still in flight (stream order covers every later use)
	if (!deviceInput) BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
	s->haveResult = true;
	return BHIP_OK;
}

static int surfCreateUnregistered(bhip_ctx* ctx, const bhip_fh_cfg* fh, const bhip_surf_cfg* surf, const bhip_ori_cfg* ori, int stable, bhip_surf** out);

// device buffer growth that keeps the first `keep` bytes (result arrays that chunks are appended to)
static int growKeep(bhip_ctx* ctx, DevBuf& b, size_t bytes, size_t keep) {
	if (bytes <= b.cap) return BHIP_OK;
	DevBuf n;
	BHIP_TRY(n.reserve(ctx, bytes + bytes / 4));
	if (b.p && keep) BHIP_HIP(ctx, hipMemcpyAsync(n.p, b.p, keep, hipMemcpyDeviceToDevice, ctx->stream));
	if (b.p) { BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream)); b.release(); }
	b = n;
	return BHIP_OK;
}

#define SURF_CHUNK 32   // frames per chunk of a host batch (265 MB of 1080p GrayF32: ~5 ms of PCIe, ~3 ms of kernels)

// Host frames -> device in chunks on the copy stream, each chunk detected + described by the worker object as soon as it has arrived;
// results are appended to the owner's arrays, which end up exactly as one surfRun over the whole batch leaves them.
// upload(i, dst, stream) enqueues the copy of frame i.  Returns BHIP_OK with *done = false when the batch has to go through the plain
// path (small batch, or a key-point list outgrew the owner's capacity half way).
template <class Upload>
static int surfRunChunked(bhip_surf* s, int width, int height, int batch, size_t imgBytes, bool u8, Upload upload, bool* done) {
	bhip_ctx* ctx = s->ctx;
	*done = false;
	int chunk = SURF_CHUNK;
	{ const char* e = getenv("BHIP_SURF_CHUNK"); if (e && atoi(e) > 0) chunk = atoi(e); }   // tests: chunk small batches too
	if (batch < 2 * chunk) return BHIP_OK;
	if (!s->worker) {
		BHIP_TRY(surfCreateUnregistered(ctx, &s->det.cfg, &s->sd, &s->ori, s->stable, &s->worker));
		if (s->brief) {
			// the worker samples with the owner's definition (it borrows the device table; briefTab stays empty so it is never freed twice)
			bhip_surf* w0 = s->worker;
			w0->brief = true; w0->briefRadius = s->briefRadius; w0->briefPoints = s->briefPoints; w0->briefWords = s->briefWords;
			w0->briefBorrow = s->briefTab.as<int>(); w0->briefCompareOff = s->briefCompareOff; w0->briefPatch = s->briefPatch;
		}
		BHIP_HIP(ctx, hipStreamCreateWithFlags(&s->copyStream, hipStreamNonBlocking));
	}
	bhip_surf* w = s->worker;
	const size_t px = (size_t)width * height;
	s->haveResult = false;
	s->det.intTaps = u8;
	BHIP_TRY(s->det.prepare(ctx, width, height, batch));
	BHIP_TRY(s->iiBuf.reserve(ctx, px * 4 * batch));
	// everything queued on the compute stream so far (an earlier batch may still read inBuf) precedes the first upload
	const int nchunks = (batch + chunk - 1) / chunk;
	// events are destroyed on every way out of this function
	struct EventSet {
		std::vector<hipEvent_t> evs;
		~EventSet() { for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e); }
	} events;
	events.evs.assign(nchunks + 1, nullptr);
	hipEvent_t& ev = events.evs[nchunks];
	hipEvent_t* arrived = events.evs.data();
	BHIP_HIP(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
	BHIP_HIP(ctx, hipEventRecord(ev, ctx->stream));
	BHIP_HIP(ctx, hipStreamWaitEvent(s->copyStream, ev, 0));
	int status = BHIP_OK;
	for (int c = 0; c < nchunks && status == BHIP_OK; c++) {
		const int a = c * chunk, n = std::min(chunk, batch - a);
		for (int i = a; i < a + n && status == BHIP_OK; i++) status = upload(i, (char*)s->inBuf.p + imgBytes * i, s->copyStream);
		if (status == BHIP_OK && (hipEventCreateWithFlags(&arrived[c], hipEventDisableTiming) != hipSuccess || hipEventRecord(arrived[c], s->copyStream) != hipSuccess))
			status = bhip_fail(ctx, BHIP_ERR_HIP, "event");
	}
	const int dof = s->tables.dof;
	s->starts.assign(batch + 1, 0);
	s->det.counts.assign(batch, 0);
	long long total = 0;
	bool fits = true;
	for (int c = 0; c < nchunks && status == BHIP_OK && fits; c++) {
		const int a = c * chunk, n = std::min(chunk, batch - a);
		if (hipStreamWaitEvent(ctx->stream, arrived[c], 0) != hipSuccess) { status = bhip_fail(ctx, BHIP_ERR_HIP, "wait"); break; }
		w->extII = s->iiBuf.as<float>() + px * a;
		ImgView in{(const float*)((const char*)s->inBuf.p + imgBytes * a), (long long)px, width, width, height};   // u8: only pointer and shape are used
		status = surfRun(w, in, n, 0, u8);
		if (status != BHIP_OK) break;
		int maxCount = 0;
		for (int i = 0; i < n; i++) maxCount = std::max(maxCount, w->det.counts[i]);
		if (maxCount > s->det.cap) { fits = false; break; }   // the owner's [image][cap] key-point array is too narrow: plain path (it grows there)
		const long long ct = w->det.total;
		status = growKeep(ctx, s->angBuf, (size_t)std::max<long long>(total + ct, 1) * 8, (size_t)total * 8);
		if (status == BHIP_OK && !s->brief) status = growKeep(ctx, s->descBuf, (size_t)std::max<long long>(total + ct, 1) * 8 * dof, (size_t)total * 8 * dof);
		if (status == BHIP_OK) status = growKeep(ctx, s->whiteBuf, (size_t)std::max<long long>(total + ct, 1), (size_t)total);
		if (status == BHIP_OK && s->brief) status = growKeep(ctx, s->wordsBuf, (size_t)std::max<long long>(total + ct, 1) * 4 * s->briefWords, (size_t)total * 4 * s->briefWords);
		if (status != BHIP_OK) break;
		hipError_t e = hipSuccess;
		if (maxCount > 0)
			e = hipMemcpy2DAsync(s->det.sorted.as<KeyPoint>() + (long long)a * s->det.cap, (size_t)s->det.cap * sizeof(KeyPoint), w->det.sorted.p,
								 (size_t)w->det.cap * sizeof(KeyPoint), (size_t)maxCount * sizeof(KeyPoint), n, hipMemcpyDeviceToDevice, ctx->stream);
		if (e == hipSuccess && ct > 0) e = hipMemcpyAsync(s->angBuf.as<double>() + total, w->angBuf.p, (size_t)ct * 8, hipMemcpyDeviceToDevice, ctx->stream);
		if (e == hipSuccess && ct > 0 && !s->brief) e = hipMemcpyAsync(s->descBuf.as<double>() + total * dof, w->descBuf.p, (size_t)ct * 8 * dof, hipMemcpyDeviceToDevice, ctx->stream);
		if (e == hipSuccess && ct > 0 && s->brief) e = hipMemcpyAsync(s->wordsBuf.as<int>() + total * s->briefWords, w->wordsBuf.p, (size_t)ct * 4 * s->briefWords, hipMemcpyDeviceToDevice, ctx->stream);
		if (e == hipSuccess && ct > 0) e = hipMemcpyAsync(s->whiteBuf.as<uint8_t>() + total, w->whiteBuf.p, (size_t)ct, hipMemcpyDeviceToDevice, ctx->stream);
		if (e != hipSuccess) { status = bhip_fail(ctx, BHIP_ERR_HIP, hipGetErrorString(e)); break; }
		for (int i = 0; i < n; i++) {
			s->det.counts[a + i] = w->det.counts[i];
			s->starts[a + i + 1] = s->starts[a + i] + w->det.counts[i];
		}
		total += ct;
	}
	// every upload has to be over before the caller's frames may change (and before the plain path re-uses inBuf)
	(void)hipStreamSynchronize(s->copyStream);
	(void)hipStreamSynchronize(ctx->stream);
	w->extII = nullptr;
	if (status != BHIP_OK) return status;
	if (!fits) {
		// the frames are all on the device: run the whole batch the plain way (its detector grows the key-point capacity as needed)
		ImgView in{s->inBuf.as<float>(), (long long)px, width, width, height};
		*done = true;
		return surfRun(s, in, batch, 0, u8);
	}
	if (total > 0x7fffffffLL) return bhip_fail(ctx, BHIP_ERR_CAPACITY, "more than 2^31 key points in one batch");
	s->det.total = total;
	s->W = width; s->H = height; s->batch = batch;
	s->iiView = ImgView{s->iiBuf.as<float>(), (long long)px, width, width, height};
	s->planarBands = 0;
	s->planar = DescPlanar{s->iiBuf.as<float>() + (long long)px, (long long)px, (long long)px, 0, 2.0, u8};
	s->descOptions = u8;
	BHIP_TRY(s->startBuf.reserve(ctx, (size_t)(batch + 1) * 4));
	BHIP_HIP(ctx, hipMemcpyAsync(s->startBuf.p, s->starts.data(), (size_t)(batch + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
	BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
	s->haveResult = true;
	*done = true;
	return BHIP_OK;
}

static int surfCreateUnregistered(bhip_ctx* ctx, const bhip_fh_cfg* fh, const bhip_surf_cfg* surf, const bhip_ori_cfg* ori, int stable, bhip_surf** out) {
	CHECK_CTX(ctx);
	if (!out) return bhip_fail(ctx, BHIP_ERR_INVALID, "null output");
	*out = nullptr;
	std::unique_ptr<bhip_surf> s(new (std::nothrow) bhip_surf());
	if (!s) return bhip_fail(ctx, BHIP_ERR_NOMEM, "out of host memory");
	s->ctx = ctx;
	s->stable = stable ? 1 : 0;
	if (fh) s->det.cfg = *fh; else bhip_fh_cfg_default(&s->det.cfg);
	if (surf) s->sd = *surf; else bhip_surf_cfg_default(&s->sd);
	if (ori) s->ori = *ori; else bhip_ori_cfg_default(&s->ori, s->stable);
	const int st = buildTables(s.get());
	if (st != BHIP_OK) { s->tabBuf.release(); return st; }
	*out = s.release();
	return BHIP_OK;
}

// Frees everything s holds on the device (its context must still be alive) and detaches it from the context: s is an inert shell afterwards.
// The chunk worker is owned by s and is not in the registry.
static void surfReleaseDevice(bhip_surf* s) {
	if (!s || !s->ctx) return;
	(void)hipSetDevice(s->ctx->device);
	(void)hipStreamSynchronize(s->ctx->stream);
	if (s->worker) { surfReleaseDevice(s->worker); delete s->worker; s->worker = nullptr; }
	if (s->copyStream) { (void)hipStreamSynchronize(s->copyStream); (void)hipStreamDestroy(s->copyStream); s->copyStream = nullptr; }
	s->det.release();
	if (s->startsPinned) { (void)hipHostFree(s->startsPinned); s->startsPinned = nullptr; s->startsPinnedBytes = 0; }
	DevBuf* bufs[] = {&s->tabBuf, &s->inBuf, &s->iiBuf, &s->startBuf, &s->angBuf, &s->descBuf, &s->whiteBuf, &s->xysBuf, &s->tmpKp, &s->tmpAng, &s->tmpDesc, &s->tmpWhite, &s->permBuf,
					  &s->briefTab, &s->wordsBuf};
	for (DevBuf* b : bufs) b->release();
	s->briefBorrow = nullptr;
	s->haveResult = false;
	s->ctx = nullptr;
}
static void surfOrphanChildren(bhip_ctx* ctx) {
	for (bhip_surf* s : registry().surfs)
		if (s->ctx == ctx) surfReleaseDevice(s);
}

// every sample point the pairs use lies within [-radius, radius]^2: the LDS-patch BRIEF kernel may be used (ip.hip, k_brief_patch)
static bool briefPatchOk(const int32_t* samplePoints, int nSamples, int radius) {
	for (int i = 0; i < 2 * nSamples; i++)
		if (samplePoints[i] < -radius || samplePoints[i] > radius) return false;
	return radius >= 0 && radius <= 40;
}

extern "C" {

int bhip_surf_create(bhip_ctx* ctx, const bhip_fh_cfg* fh, const bhip_surf_cfg* surf, const bhip_ori_cfg* ori, int stable, bhip_surf** out) {
	HandleRegistry& R = registry();
	{
		std::lock_guard<std::mutex> lock(R.m);
		if (!ctx || !R.ctxs.count(ctx)) return BHIP_ERR_INVALID;
	}
	BHIP_TRY(surfCreateUnregistered(ctx, fh, surf, ori, stable, out));
	std::lock_guard<std::mutex> lock(R.m);
	R.surfs.insert(*out);
	return BHIP_OK;
}

int bhip_surf_destroy(bhip_surf* s) {
	if (!s) return BHIP_OK;
	HandleRegistry& R = registry();
	std::lock_guard<std::mutex> lock(R.m);
	if (R.exiting) return BHIP_OK;                       // process teardown: the runtime reclaims everything
	if (!R.surfs.count(s)) return BHIP_ERR_INVALID;      // not a live object (destroyed twice, or never created)
	R.surfs.erase(s);
	surfReleaseDevice(s);                                // no-op when the context went first
	delete s;
	return BHIP_OK;
}

int bhip_surf_dof(bhip_surf* s) { return s ? s->dofOut() : 0; }

int bhip_surf_detect_dev_f32(bhip_surf* s, const float* dev_images, long long imageStride, int stride, int width, int height, int batch) {
	if (!s) return BHIP_ERR_INVALID;
	bhip_ctx* ctx = s->ctx;
	CHECK_CTX(ctx);
	if (!dev_images || width <= 0 || height <= 0 || batch <= 0 || stride < width) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad image batch");
	ImgView in{dev_images, imageStride, stride, width, height};
	return surfRun(s, in, batch, 0, false, true);
}

int bhip_surf_detect_f32(bhip_surf* s, const float* const* img, const int* startIndex, const int* stride, int width, int height, int batch) {
	if (!s) return BHIP_ERR_INVALID;
	bhip_ctx* ctx = s->ctx;
	CHECK_CTX(ctx);
	if (!img || width <= 0 || height <= 0 || batch <= 0) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad image batch");
	const size_t imgBytes = (size_t)width * height * 4;
	BHIP_TRY(s->inBuf.reserve(ctx, imgBytes * batch));
	for (int i = 0; i < batch; i++)
		if (!img[i] || (stride ? stride[i] : width) < width) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad image (null or stride < width)");
	auto upload = [&](int i, void* dst, hipStream_t st_) -> int {
		const int st = stride ? stride[i] : width;
		const float* src = img[i] + (startIndex ? startIndex[i] : 0);
		// a dense frame is one linear copy (the 2-D form is several times slower over PCIe even when pitch == width)
		if (st == width) BHIP_HIP(ctx, hipMemcpyAsync(dst, src, imgBytes, hipMemcpyHostToDevice, st_));
		else BHIP_HIP(ctx, hipMemcpy2DAsync(dst, (size_t)width * 4, src, (size_t)st * 4, (size_t)width * 4, height, hipMemcpyHostToDevice, st_));
		return BHIP_OK;
	};
	bool done = false;
	BHIP_TRY(surfRunChunked(s, width, height, batch, imgBytes, false, upload, &done));
	if (done) return BHIP_OK;
	for (int i = 0; i < batch; i++) BHIP_TRY(upload(i, (char*)s->inBuf.p + imgBytes * i, ctx->stream));
	ImgView in{s->inBuf.as<float>(), (long long)width * height, width, width, height};
	return surfRun(s, in, batch);
}

// FactoryDetectDescribe.surfColorStable / surfColorFast on one Planar<GrayF32> frame:
//   SurfPlanar_to_DetectDescribePoint.detect      F:abst/feature/detdesc/SurfPlanar_to_DetectDescribePoint.java:62-77
//   ImplConvertPlanarToGray.average               I:core/image/impl/ImplConvertPlanarToGray.java:296-336
//   DetectDescribeSurfPlanar.detect / describe    F:alg/feature/detdesc/DetectDescribeSurfPlanar.java:91-124
//   DescribePointSurfPlanar.describe              F:alg/feature/describe/DescribePointSurfPlanar.java:100-114
int bhip_surf_detect_planar_f32(bhip_surf* s, const float* const* bands, int numBands, int startIndex, int stride, int width, int height) {
	if (!s) return BHIP_ERR_INVALID;
	bhip_ctx* ctx = s->ctx;
	CHECK_CTX(ctx);
	if (!bands || numBands < 1 || numBands > 16 || width <= 0 || height <= 0 || stride < width) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad planar image");
	const size_t px = (size_t)width * height;
	BHIP_TRY(s->inBuf.reserve(ctx, px * 4 * (1 + numBands)));
	for (int b = 0; b < numBands; b++) {
		if (!bands[b]) return bhip_fail(ctx, BHIP_ERR_INVALID, "null band");
		BHIP_HIP(ctx, hipMemcpy2DAsync((char*)s->inBuf.p + px * 4 * (1 + b), (size_t)width * 4, bands[b] + startIndex, (size_t)stride * 4, (size_t)width * 4, height,
									   hipMemcpyHostToDevice, ctx->stream));
	}
	BHIP_TRY(bhip_launch_planar_average(ctx, s->inBuf.as<float>() + px, (long long)px, numBands, (long long)px, s->inBuf.as<float>()));
	ImgView in{s->inBuf.as<float>(), (long long)px, width, width, height};
	return surfRun(s, in, 1, numBands);
}

// FactoryDetectDescribe.surfStable / surfFast on GrayU8 frames (integral type GrayS32, GIntegralImageOps.getIntegralType): same results
// interface as bhip_surf_detect_f32
int bhip_surf_detect_u8(bhip_surf* s, const uint8_t* const* img, const int* startIndex, const int* stride, int width, int height, int batch) {
	if (!s) return BHIP_ERR_INVALID;
	bhip_ctx* ctx = s->ctx;
	CHECK_CTX(ctx);
	if (!img || width <= 0 || height <= 0 || batch <= 0) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad image batch");
	const size_t imgBytes = (size_t)width * height;
	BHIP_TRY(s->inBuf.reserve(ctx, imgBytes * batch));
	for (int i = 0; i < batch; i++)
		if (!img[i] || (stride ? stride[i] : width) < width) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad image (null or stride < width)");
	auto upload = [&](int i, void* dst, hipStream_t st_) -> int {
		const int st = stride ? stride[i] : width;
		const uint8_t* src = img[i] + (startIndex ? startIndex[i] : 0);
		if (st == width) BHIP_HIP(ctx, hipMemcpyAsync(dst, src, imgBytes, hipMemcpyHostToDevice, st_));
		else BHIP_HIP(ctx, hipMemcpy2DAsync(dst, (size_t)width, src, (size_t)st, (size_t)width, height, hipMemcpyHostToDevice, st_));
		return BHIP_OK;
	};
	bool done = false;
	BHIP_TRY(surfRunChunked(s, width, height, batch, imgBytes, true, upload, &done));
	if (done) return BHIP_OK;
	for (int i = 0; i < batch; i++) BHIP_TRY(upload(i, (char*)s->inBuf.p + imgBytes * i, ctx->stream));
	ImgView in{s->inBuf.as<float>(), (long long)width * height, width, width, height};   // only the pointer and the shape are used
	return surfRun(s, in, batch, 0, true);
}

int bhip_surf_count(bhip_surf* s, int image, int* n) {
	if (!s || !n) return BHIP_ERR_INVALID;
	if (!s->haveResult || image < 0 || image >= s->batch) return bhip_fail(s->ctx, BHIP_ERR_INVALID, "no detect result for that image");
	*n = s->det.counts[image];
	return BHIP_OK;
}
// getNumberOfFeatures() of every image of the last detect in one call (a batch-level caller's prefix sums)
int bhip_surf_counts(bhip_surf* s, int* counts, int capacity) {
	if (!s || !counts) return BHIP_ERR_INVALID;
	if (!s->haveResult || capacity < s->batch) return bhip_fail(s->ctx, BHIP_ERR_INVALID, "no detect result / counts array too short");
	for (int i = 0; i < s->batch; i++) counts[i] = s->det.counts[i];
	return BHIP_OK;
}
int bhip_surf_total(bhip_surf* s, long long* n) {
	if (!s || !n) return BHIP_ERR_INVALID;
	*n = s->haveResult ? s->det.total : 0;
	return BHIP_OK;
}

int bhip_surf_fetch(bhip_surf* s, int image, double* xy_scale, double* angle, uint8_t* white, double* desc) {
	// (dof below is the full descriptor length: numBands * 64 after a planar detect)
	if (!s) return BHIP_ERR_INVALID;
	bhip_ctx* ctx = s->ctx;
	CHECK_CTX(ctx);
	if (!s->haveResult || image < 0 || image >= s->batch) return bhip_fail(ctx, BHIP_ERR_INVALID, "no detect result for that image");
	if (desc && s->brief) return bhip_fail(ctx, BHIP_ERR_INVALID, "this object describes with BRIEF: fetch the words with bhip_surf_fetch_brief");
	const int n = s->det.counts[image];
	if (n == 0) return BHIP_OK;
	const long long off = s->starts[image];
	const int dof = s->dofOut();
	std::vector<KeyPoint> kps;
	if (xy_scale) {
		kps.resize(n);
		BHIP_HIP(ctx, hipMemcpyAsync(kps.data(), s->det.sorted.as<KeyPoint>() + (long long)image * s->det.cap, (size_t)n * sizeof(KeyPoint),
									 hipMemcpyDeviceToHost, ctx->stream));
	}
	if (angle) BHIP_HIP(ctx, hipMemcpyAsync(angle, s->angBuf.as<double>() + off, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
	if (white) BHIP_HIP(ctx, hipMemcpyAsync(white, s->whiteBuf.as<uint8_t>() + off, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
	if (desc) BHIP_HIP(ctx, hipMemcpyAsync(desc, s->descBuf.as<double>() + off * dof, (size_t)n * 8 * dof, hipMemcpyDeviceToHost, ctx->stream));
	BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
	if (xy_scale)
		for (int i = 0; i < n; i++) { xy_scale[3 * i] = kps[i].x; xy_scale[3 * i + 1] = kps[i].y; xy_scale[3 * i + 2] = kps[i].scale; }
	return BHIP_OK;
}

// the whole batch of the last detect in ONE set of copies (per-image slices start at the exclusive prefix of bhip_surf_count):
// what a batched caller (DetectDescribeSurfHip.detectBatch) uses instead of `batch` bhip_surf_fetch calls
int bhip_surf_fetch_all(bhip_surf* s, double* xy_scale, double* angle, uint8_t* white, double* desc) {
	if (!s) return BHIP_ERR_INVALID;
	bhip_ctx* ctx = s->ctx;
	CHECK_CTX(ctx);
	if (!s->haveResult) return bhip_fail(ctx, BHIP_ERR_INVALID, "no detect result");
	if (desc && s->brief) return bhip_fail(ctx, BHIP_ERR_INVALID, "this object describes with BRIEF: fetch the words with bhip_surf_fetch_brief");
	const long long total = s->det.total;
	if (total == 0) return BHIP_OK;
	const int dof = s->dofOut();
	if (xy_scale) {
		// key points sit in [image][cap] KeyPoint records: pack x, y, scale of every image into the compact layout on the device first
		BHIP_TRY(s->xysBuf.reserve(ctx, (size_t)total * 24));
		for (int i = 0; i < s->batch; i++) {
			const int n = s->det.counts[i];
			if (n > 0)
				BHIP_HIP(ctx, hipMemcpy2DAsync((char*)s->xysBuf.p + (size_t)s->starts[i] * 24, 24, s->det.sorted.as<KeyPoint>() + (long long)i * s->det.cap,
											   sizeof(KeyPoint), 24, (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
		}
		BHIP_HIP(ctx, hipMemcpyAsync(xy_scale, s->xysBuf.p, (size_t)total * 24, hipMemcpyDeviceToHost, ctx->stream));
	}
	if (angle) BHIP_HIP(ctx, hipMemcpyAsync(angle, s->angBuf.p, (size_t)total * 8, hipMemcpyDeviceToHost, ctx->stream));
	if (white) BHIP_HIP(ctx, hipMemcpyAsync(white, s->whiteBuf.p, (size_t)total, hipMemcpyDeviceToHost, ctx->stream));
	if (desc) BHIP_HIP(ctx, hipMemcpyAsync(desc, s->descBuf.p, (size_t)total * 8 * dof, hipMemcpyDeviceToHost, ctx->stream));
	BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return BHIP_OK;
}

int bhip_surf_dev_view(bhip_surf* s, int image, const double** dev_desc, const double** dev_xy_scale, const uint8_t** dev_white, int* n) {
	if (!s) return BHIP_ERR_INVALID;
	if (!s->haveResult || image < 0 || image >= s->batch) return bhip_fail(s->ctx, BHIP_ERR_INVALID, "no detect result for that image");
	const long long off = s->starts[image];
	if (dev_desc) *dev_desc = s->descBuf.as<double>() + off * s->dofOut();
	// key points are KeyPoint records {x,y,scale,key,pad}: 32-byte stride, first three doubles are x,y,scale
	if (dev_xy_scale) *dev_xy_scale = (const double*)(s->det.sorted.as<KeyPoint>() + (long long)image * s->det.cap);
	if (dev_white) *dev_white = s->whiteBuf.as<uint8_t>() + off;
	if (n) *n = s->det.counts[image];
	return BHIP_OK;
}

// FactoryDetectDescribe.fuseTogether(FactoryInterestPoint.fastHessian(fh), null, FactoryDescribeRegionPoint.brief(config, imageType))
// (F:factory/feature/detdesc/FactoryDetectDescribe.java:279-284, F:factory/feature/describe/FactoryDescribeRegionPoint.java:187-202 with
// config.fixed): the returned object is driven through the same bhip_surf_detect_* / _count / _fetch calls as the SURF one.
int bhip_surf_create_brief(bhip_ctx* ctx, const bhip_fh_cfg* fh, int radius, int numPoints, const int32_t* samplePoints, const int32_t* compare, bhip_surf** out) {
	HandleRegistry& R = registry();
	{
		std::lock_guard<std::mutex> lock(R.m);
		if (!ctx || !R.ctxs.count(ctx)) return BHIP_ERR_INVALID;
	}
	CHECK_CTX(ctx);
	if (!out) return bhip_fail(ctx, BHIP_ERR_INVALID, "null output");
	*out = nullptr;
	if (radius < 0 || numPoints <= 0 || !samplePoints || !compare) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad BRIEF definition");
	int maxIdx = 0;
	for (int i = 0; i < 2 * numPoints; i++) {
		if (compare[i] < 0) return bhip_fail(ctx, BHIP_ERR_INVALID, "negative sample index");
		maxIdx = std::max(maxIdx, compare[i]);
	}
	bhip_surf* s = nullptr;
	BHIP_TRY(surfCreateUnregistered(ctx, fh, nullptr, nullptr, 0, &s));   // (the SURF tables of the shell are never used)
	s->brief = true;
	s->briefRadius = radius; s->briefPoints = numPoints; s->briefWords = (numPoints + 31) / 32;
	const size_t nSample = (size_t)(maxIdx + 1) * 2, nCompare = (size_t)numPoints * 2;
	s->briefCompareOff = nSample;
	s->briefPatch = briefPatchOk(samplePoints, maxIdx + 1, radius);
	int st = s->briefTab.reserve(ctx, (nSample + nCompare) * 4);
	if (st == BHIP_OK && hipMemcpy(s->briefTab.p, samplePoints, nSample * 4, hipMemcpyHostToDevice) != hipSuccess) st = bhip_fail(ctx, BHIP_ERR_HIP, "BRIEF table upload");
	if (st == BHIP_OK && hipMemcpy(s->briefTab.as<int>() + nSample, compare, nCompare * 4, hipMemcpyHostToDevice) != hipSuccess) st = bhip_fail(ctx, BHIP_ERR_HIP, "BRIEF table upload");
	if (st != BHIP_OK) { surfReleaseDevice(s); delete s; return st; }
	std::lock_guard<std::mutex> lock(R.m);
	R.surfs.insert(s);
	*out = s;
	return BHIP_OK;
}

// getDescription(i).data of every feature of one image (image >= 0: count * words ints) or of the whole batch (image = -1: total * words
// ints, image i's slice at the exclusive prefix of the counts); words = ceil(numPoints / 32) (TupleDesc_B, T:struct/feature/TupleDesc_B.java)
int bhip_surf_fetch_brief(bhip_surf* s, int image, int32_t* words) {
	if (!s || !words) return BHIP_ERR_INVALID;
	bhip_ctx* ctx = s->ctx;
	CHECK_CTX(ctx);
	if (!s->brief) return bhip_fail(ctx, BHIP_ERR_INVALID, "not a BRIEF detect+describe object");
	if (!s->haveResult || image < -1 || image >= s->batch) return bhip_fail(ctx, BHIP_ERR_INVALID, "no detect result for that image");
	const long long off = image < 0 ? 0 : s->starts[image];
	const long long n = image < 0 ? s->det.total : s->det.counts[image];
	if (n == 0) return BHIP_OK;
	BHIP_HIP(ctx, hipMemcpyAsync(words, s->wordsBuf.as<int>() + off * s->briefWords, (size_t)n * 4 * s->briefWords, hipMemcpyDeviceToHost, ctx->stream));
	BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return BHIP_OK;
}
int bhip_surf_dev_view_brief(bhip_surf* s, int image, const int32_t** dev_words, int* words, int* n) {
	if (!s) return BHIP_ERR_INVALID;
	if (!s->brief) return bhip_fail(s->ctx, BHIP_ERR_INVALID, "not a BRIEF detect+describe object");
	if (!s->haveResult || image < 0 || image >= s->batch) return bhip_fail(s->ctx, BHIP_ERR_INVALID, "no detect result for that image");
	if (dev_words) *dev_words = s->wordsBuf.as<int>() + (long long)s->starts[image] * s->briefWords;
	if (words) *words = s->briefWords;
	if (n) *n = s->det.counts[image];
	return BHIP_OK;
}

int bhip_surf_fetch_integral(bhip_surf* s, int image, float* out) {
	if (!s || !out) return BHIP_ERR_INVALID;
	bhip_ctx* ctx = s->ctx;
	CHECK_CTX(ctx);
	if (!s->haveResult || image < 0 || image >= s->batch) return bhip_fail(ctx, BHIP_ERR_INVALID, "no detect result for that image");
	BHIP_HIP(ctx, hipMemcpyAsync(out, s->iiBuf.as<float>() + (long long)image * s->W * s->H, (size_t)s->W * s->H * 4, hipMemcpyDeviceToHost, ctx->stream));
	BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return BHIP_OK;
}

int bhip_surf_describe_points(bhip_surf* s, int image, const double* xy_scale, int n, double* angle, uint8_t* white, double* desc) {
	if (!s) return BHIP_ERR_INVALID;
	bhip_ctx* ctx = s->ctx;
	CHECK_CTX(ctx);
	if (!s->haveResult || image < 0 || image >= s->batch) return bhip_fail(ctx, BHIP_ERR_INVALID, "no detect result for that image");
	if (s->brief) return bhip_fail(ctx, BHIP_ERR_INVALID, "this object describes with BRIEF");
	if (n < 0 || (n > 0 && !xy_scale)) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad point list");
	if (n == 0) return BHIP_OK;
	std::vector<KeyPoint> kps(n);
	for (int i = 0; i < n; i++) { kps[i].x = xy_scale[3 * i]; kps[i].y = xy_scale[3 * i + 1]; kps[i].scale = xy_scale[3 * i + 2]; kps[i].key = 0; kps[i].pad = 0; }
	const int dof = s->dofOut();
	BHIP_TRY(s->tmpKp.reserve(ctx, (size_t)n * sizeof(KeyPoint)));
	BHIP_TRY(s->tmpAng.reserve(ctx, (size_t)n * 8));
	BHIP_TRY(s->tmpDesc.reserve(ctx, (size_t)n * 8 * dof));
	BHIP_TRY(s->tmpWhite.reserve(ctx, (size_t)n));
	BHIP_HIP(ctx, hipMemcpyAsync(s->tmpKp.p, kps.data(), (size_t)n * sizeof(KeyPoint), hipMemcpyHostToDevice, ctx->stream));
	BHIP_TRY(bhip_launch_describe_ex(ctx, s->iiView, s->tmpKp.as<KeyPoint>(), 0, nullptr, s->batch, image, n, s->tables, nullptr, s->tmpAng.as<double>(),
									 s->tmpDesc.as<double>(), s->tmpWhite.as<uint8_t>(), nullptr, s->descOptions ? &s->planar : nullptr));
	if (angle) BHIP_HIP(ctx, hipMemcpyAsync(angle, s->tmpAng.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream));
	if (white) BHIP_HIP(ctx, hipMemcpyAsync(white, s->tmpWhite.p, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
	if (desc) BHIP_HIP(ctx, hipMemcpyAsync(desc, s->tmpDesc.p, (size_t)n * 8 * dof, hipMemcpyDeviceToHost, ctx->stream));
	BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return BHIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// stage-level entry points (host buffers in / out)
// ---------------------------------------------------------------------------------------------------------------
// pitch = floats between rows of the device copy (0: dense).  pitch4(w) keeps rows 16-byte aligned so the tiled ip kernels apply.
static inline int pitch4(int w) { return (w + 3) & ~3; }
static int uploadImage(bhip_ctx* ctx, DevBuf& buf, const float* in, int start, int stride, int w, int h, int pitch = 0) {
	if (pitch == 0) pitch = w;
	BHIP_TRY(buf.reserve(ctx, (size_t)pitch * h * 4));
	BHIP_HIP(ctx, hipMemcpy2DAsync(buf.p, (size_t)pitch * 4, in + start, (size_t)stride * 4, (size_t)w * 4, h, hipMemcpyHostToDevice, ctx->stream));
	return BHIP_OK;
}
static int downloadImage(bhip_ctx* ctx, const void* dev, float* out, int start, int stride, int w, int h, int pitch = 0) {
	if (pitch == 0) pitch = w;
	BHIP_HIP(ctx, hipMemcpy2DAsync(out + start, (size_t)stride * 4, dev, (size_t)pitch * 4, (size_t)w * 4, h, hipMemcpyDeviceToHost, ctx->stream));
	BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return BHIP_OK;
}
#define CHECK_IMG(ctx, p, stride, w, h)                                                                                 \
	do {                                                                                                                \
		if (!(p) || (w) <= 0 || (h) <= 0 || (stride) < (w)) return bhip_fail((ctx), BHIP_ERR_INVALID, "bad image");     \
	} while (0)

int bhip_integral_f32(bhip_ctx* ctx, const float* in, int inStart, int inStride, int width, int height, float* out, int outStart, int outStride) {
	CHECK_CTX(ctx);
	CHECK_IMG(ctx, in, inStride, width, height);
	CHECK_IMG(ctx, out, outStride, width, height);
	CtxScratch* sc = scratchOf(ctx);
	BHIP_TRY(uploadImage(ctx, sc->a, in, inStart, inStride, width, height));
	BHIP_TRY(sc->b.reserve(ctx, (size_t)width * height * 4));
	ImgView iv{sc->a.as<float>(), (long long)width * height, width, width, height};
	ImgViewW ov{sc->b.as<float>(), (long long)width * height, width, width, height};
	BHIP_TRY(bhip_launch_integral(ctx, iv, ov, 1));
	return downloadImage(ctx, sc->b.p, out, outStart, outStride, width, height);
}

int bhip_hessian_f32(bhip_ctx* ctx, const float* ii, int iiStart, int iiStride, int width, int height, int skip, int size, float* intensity,
					 int outStart, int outStride) {
	CHECK_CTX(ctx);
	CHECK_IMG(ctx, ii, iiStride, width, height);
	if (skip < 1 || size < 3) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad skip/size");
	const int w = width / skip, h = height / skip;
	if (w <= 0 || h <= 0) return BHIP_OK;
	CHECK_IMG(ctx, intensity, outStride, w, h);
	CtxScratch* sc = scratchOf(ctx);
	BHIP_TRY(uploadImage(ctx, sc->a, ii, iiStart, iiStride, width, height));
	BHIP_TRY(sc->b.reserve(ctx, (size_t)w * h * 4));
	ImgView iv{sc->a.as<float>(), (long long)width * height, width, width, height};
	BHIP_TRY(bhip_launch_hessian(ctx, iv, 1, skip, 1, &size, sc->b.as<float>(), (long long)w * h, (long long)w * h, w));
	return downloadImage(ctx, sc->b.p, intensity, outStart, outStride, w, h);
}

// strict block NMS of `batch` dense device images: lists into dev_xy ([batch][cap] (x,y) int16 pairs, block-raster order), counts into
// dev_n[batch] (a count may exceed cap: only the first cap pairs are written)
static int nonmaxDevice(bhip_ctx* ctx, const float* dev_intensity, long long imageStride, int stride, int width, int height, int batch, int radius,
						float threshold, int border, int16_t* dev_xy, int cap, int* dev_n) {
	if (radius < 1) return bhip_fail(ctx, BHIP_ERR_INVALID, "Search radius must be >= 1");
	if (border < 0) return bhip_fail(ctx, BHIP_ERR_INVALID, "Ignore border must be >= 0 ");
	if (width >= 32768 || height >= 32768) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "image too large for Point2D_I16");
	BHIP_HIP(ctx, hipMemsetAsync(dev_n, 0, (size_t)batch * 4, ctx->stream));
	const int step = radius + 1;
	const int rw = width - 2 * border, rh = height - 2 * border;
	if (rw <= 0 || rh <= 0) return BHIP_OK;
	const int nbx = (rw + step - 1) / step, nby = (rh + step - 1) / step;
	const int words = (int)(((long long)nbx * nby + 31) / 32) + 1;
	CtxScratch* sc = scratchOf(ctx);
	BHIP_TRY(sc->nmsBitmap.reserve(ctx, (size_t)words * 4 * batch));
	BHIP_TRY(sc->nmsPrefix.reserve(ctx, (size_t)words * 4 * batch));
	BHIP_TRY(sc->nmsPos.reserve(ctx, (size_t)nbx * nby * 2 * batch));
	BHIP_HIP(ctx, hipMemsetAsync(sc->nmsBitmap.p, 0, (size_t)words * 4 * batch, ctx->stream));
	BHIP_TRY(bhip_launch_nonmax_blocks(ctx, dev_intensity, imageStride, stride, width, height, batch, radius, threshold, border, sc->nmsBitmap.as<unsigned int>(),
									   words, sc->nmsPos.as<unsigned short>(), nbx, nby));
	BHIP_TRY(bhip_launch_word_prefix(ctx, sc->nmsBitmap.as<unsigned int>(), words, batch, sc->nmsPrefix.as<unsigned int>(), dev_n));
	BHIP_TRY(bhip_launch_blocks_to_xy(ctx, sc->nmsBitmap.as<unsigned int>(), sc->nmsPrefix.as<unsigned int>(), words, sc->nmsPos.as<unsigned short>(), nbx, nby,
									  batch, radius, border, dev_xy, cap));
	return BHIP_OK;
}

int bhip_nonmax_block_dev_f32(bhip_ctx* ctx, const float* dev_intensity, long long imageStride, int stride, int width, int height, int batch, int radius,
							  float threshold, int border, int16_t* dev_xy, int cap, int* dev_n) {
	CHECK_CTX(ctx);
	if (!dev_intensity || width <= 0 || height <= 0 || batch <= 0 || stride < width) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad image batch");
	if (!dev_n || cap < 0 || (cap > 0 && !dev_xy)) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad output");
	return nonmaxDevice(ctx, dev_intensity, imageStride, stride, width, height, batch, radius, threshold, border, dev_xy, cap, dev_n);
}

int bhip_nonmax_block_f32(bhip_ctx* ctx, const float* intensity, int start, int stride, int width, int height, int radius, float threshold,
						  int border, int16_t* xy, int cap, int* n) {
	CHECK_CTX(ctx);
	CHECK_IMG(ctx, intensity, stride, width, height);
	if (!n || cap < 0 || (cap > 0 && !xy)) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad output");
	*n = 0;
	CtxScratch* sc = scratchOf(ctx);
	BHIP_TRY(uploadImage(ctx, sc->a, intensity, start, stride, width, height));
	BHIP_TRY(sc->d.reserve(ctx, 16));
	BHIP_TRY(sc->e.reserve(ctx, (size_t)std::max(cap, 1) * 4));
	BHIP_TRY(nonmaxDevice(ctx, sc->a.as<float>(), 0, width, width, height, 1, radius, threshold, border, sc->e.as<int16_t>(), cap, sc->d.as<int>()));
	BHIP_HIP(ctx, hipMemcpyAsync(ctx->hostScratch, sc->d.p, 4, hipMemcpyDeviceToHost, ctx->stream));
	BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
	*n = ctx->hostScratch[0];
	const int ncopy = std::min(*n, cap);
	if (ncopy > 0) {
		BHIP_HIP(ctx, hipMemcpyAsync(xy, sc->e.p, (size_t)ncopy * 4, hipMemcpyDeviceToHost, ctx->stream));
		BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
	}
	return BHIP_OK;
}

int bhip_select_nbest_f32(bhip_ctx* ctx, const float* intensity, int start, int stride, int width, int height, const int16_t* xy, int n, int target,
						  int positive, int16_t* out_xy, int* out_n) {
	CHECK_CTX(ctx);
	CHECK_IMG(ctx, intensity, stride, width, height);
	if (n < 0 || !out_n || (n > 0 && (!xy || !out_xy))) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad corner list");
	for (int i = 0; i < n; i++)
		if (xy[2 * i] < 0 || xy[2 * i] >= width || xy[2 * i + 1] < 0 || xy[2 * i + 1] >= height)
			return bhip_fail(ctx, BHIP_ERR_INVALID, "corner outside the intensity image");   // GrayF32.get would throw ImageAccessException
	if (n <= target) {
		// SelectNBestFeatures.java:54-60: already few enough, an unpruned copy in the original order
		if (n > 0) memcpy(out_xy, xy, (size_t)n * 4);
		*out_n = n;
		return BHIP_OK;
	}
	*out_n = 0;
	if (target <= 0) return BHIP_OK;   // n > target, nothing to keep (QuickSelect with k = 0 is never reached with a positive N in the reference)
	CtxScratch* sc = scratchOf(ctx);
	BHIP_TRY(uploadImage(ctx, sc->a, intensity, start, stride, width, height));
	BHIP_TRY(sc->b.reserve(ctx, (size_t)n * 4));
	BHIP_TRY(sc->c.reserve(ctx, (size_t)n * 4));
	BHIP_TRY(sc->d.reserve(ctx, (size_t)n * 4));
	BHIP_TRY(sc->e.reserve(ctx, (size_t)target * 4));
	BHIP_HIP(ctx, hipMemcpyAsync(sc->b.p, xy, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
	BHIP_TRY(bhip_launch_select_nbest_xy(ctx, sc->a.as<float>(), width, sc->b.as<int16_t>(), n, target, positive != 0, sc->c.as<float>(), sc->d.as<int>(),
										 sc->e.as<int16_t>()));
	BHIP_HIP(ctx, hipMemcpyAsync(out_xy, sc->e.p, (size_t)target * 4, hipMemcpyDeviceToHost, ctx->stream));
	BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
	*out_n = target;
	return BHIP_OK;
}

int bhip_fh_detect_f32(bhip_ctx* ctx, const bhip_fh_cfg* cfg, const float* ii, int iiStart, int iiStride, int width, int height, double* xy_scale,
					   int cap, int* n) {
	CHECK_CTX(ctx);
	CHECK_IMG(ctx, ii, iiStride, width, height);
	if (!n || cap < 0 || (cap > 0 && !xy_scale)) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad output");
	*n = 0;
	FhDetector det;
	if (cfg) det.cfg = *cfg; else bhip_fh_cfg_default(&det.cfg);
	CtxScratch* sc = scratchOf(ctx);
	int status = uploadImage(ctx, sc->a, ii, iiStart, iiStride, width, height);
	if (status == BHIP_OK) status = det.prepare(ctx, width, height, 1);
	ImgView iv{sc->a.as<float>(), (long long)width * height, width, width, height};
	if (status == BHIP_OK) status = det.run(ctx, iv);
	if (status == BHIP_OK) {
		*n = det.counts[0];
		const int ncopy = std::min(*n, cap);
		if (ncopy > 0) {
			std::vector<KeyPoint> kps(ncopy);
			hipError_t e = hipMemcpyAsync(kps.data(), det.sorted.p, (size_t)ncopy * sizeof(KeyPoint), hipMemcpyDeviceToHost, ctx->stream);
			if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
			if (e != hipSuccess) status = bhip_fail(ctx, BHIP_ERR_HIP, hipGetErrorString(e));
			else for (int i = 0; i < ncopy; i++) { xy_scale[3 * i] = kps[i].x; xy_scale[3 * i + 1] = kps[i].y; xy_scale[3 * i + 2] = kps[i].scale; }
		}
	}
	(void)hipStreamSynchronize(ctx->stream);
	det.release();
	return status;
}

// FastHessianFeatureDetector<GrayS32>.detect(integral) (the integral image of a GrayU8 frame)
int bhip_fh_detect_s32(bhip_ctx* ctx, const bhip_fh_cfg* cfg, const int32_t* ii, int iiStart, int iiStride, int width, int height, double* xy_scale,
					   int cap, int* n) {
	CHECK_CTX(ctx);
	if (!ii || width <= 0 || height <= 0 || iiStride < width) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad image");
	if (!n || cap < 0 || (cap > 0 && !xy_scale)) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad output");
	*n = 0;
	FhDetector det;
	det.intTaps = true;
	if (cfg) det.cfg = *cfg; else bhip_fh_cfg_default(&det.cfg);
	CtxScratch* sc = scratchOf(ctx);
	int status = uploadImage(ctx, sc->a, (const float*)ii, iiStart, iiStride, width, height);   // 32-bit words either way
	if (status == BHIP_OK) status = det.prepare(ctx, width, height, 1);
	ImgView iv{sc->a.as<float>(), (long long)width * height, width, width, height};
	if (status == BHIP_OK) status = det.run(ctx, iv);
	if (status == BHIP_OK) {
		*n = det.counts[0];
		const int ncopy = std::min(*n, cap);
		if (ncopy > 0) {
			std::vector<KeyPoint> kps(ncopy);
			hipError_t e = hipMemcpyAsync(kps.data(), det.sorted.p, (size_t)ncopy * sizeof(KeyPoint), hipMemcpyDeviceToHost, ctx->stream);
			if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
			if (e != hipSuccess) status = bhip_fail(ctx, BHIP_ERR_HIP, hipGetErrorString(e));
			else for (int i = 0; i < ncopy; i++) { xy_scale[3 * i] = kps[i].x; xy_scale[3 * i + 1] = kps[i].y; xy_scale[3 * i + 2] = kps[i].scale; }
		}
	}
	(void)hipStreamSynchronize(ctx->stream);
	det.release();
	return status;
}

// ---------------------------------------------------------------------------------------------------------------
// association
// ---------------------------------------------------------------------------------------------------------------
int bhip_assoc_coltop_bytes(void) { return bhip_assoc_coltop_size(); }

static bool assocExactOnly(bhip_ctx* ctx) {
	CtxScratch* sc = scratchOf(ctx);
	if (sc->assocExactOnly < 0) {
		const char* e = getenv("BHIP_ASSOC_EXACT");
		sc->assocExactOnly = (e && e[0] == '1') ? 1 : 0;
	}
	return sc->assocExactOnly == 1;
}

static int assocL2Exact(bhip_ctx* ctx, const double* dev_src, int ns, const double* dev_dst, int nd, int dof, double maxErr, int backwards,
						int sqrtScore, int* dev_pairs, double* dev_fit);

int bhip_assoc_l2_dev(bhip_ctx* ctx, const double* dev_src, int ns, const double* dev_dst, int nd, int dof, double maxErr, int backwards,
					  int sqrtScore, int* dev_pairs, double* dev_fit) {
	CHECK_CTX(ctx);
	if (ns < 0 || nd < 0 || dof <= 0) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad sizes");
	if (ns == 0) return BHIP_OK;
	if (dof == 64 && !sqrtScore && nd > 0 && !assocExactOnly(ctx)) {
		// matrix-core path (fp32 MFMA candidates + exact fp64 re-score); falls through on degenerate inputs
		const long long zero = 0;
		int used = 0;
		BHIP_TRY(bhip_assoc_l2_mfma_batched(ctx, scratchOf(ctx)->mfma, dev_src, dev_dst, 1, &zero, &ns, &zero, &nd, maxErr, backwards, dev_pairs, dev_fit,
											&used));
		if (used) return BHIP_OK;
	}
	return assocL2Exact(ctx, dev_src, ns, dev_dst, nd, dof, maxErr, backwards, sqrtScore, dev_pairs, dev_fit);
}

static int assocL2Exact(bhip_ctx* ctx, const double* dev_src, int ns, const double* dev_dst, int nd, int dof, double maxErr, int backwards,
						int sqrtScore, int* dev_pairs, double* dev_fit) {
	CtxScratch* sc = scratchOf(ctx);
	void* col = nullptr;
	if (backwards && nd > 0) { BHIP_TRY(sc->d.reserve(ctx, (size_t)nd * bhip_assoc_coltop_size())); col = sc->d.p; }
	BHIP_TRY(bhip_assoc_phase1_l2(ctx, dev_src, ns, 0, dev_dst, nd, dof, maxErr, sqrtScore, dev_pairs, dev_fit, col, sc->work));
	if (col) BHIP_TRY(bhip_assoc_phase2(ctx, col, 1, nd, ns, 0, dev_pairs, dev_fit));
	return BHIP_OK;
}
int bhip_assoc_hamming_dev(bhip_ctx* ctx, const int32_t* dev_src, int ns, const int32_t* dev_dst, int nd, int words, double maxErr, int backwards,
						   int* dev_pairs, double* dev_fit) {
	CHECK_CTX(ctx);
	if (ns < 0 || nd < 0 || words <= 0) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad sizes");
	if (ns == 0) return BHIP_OK;
	CtxScratch* sc = scratchOf(ctx);
	void* col = nullptr;
	if (backwards && nd > 0) { BHIP_TRY(sc->d.reserve(ctx, (size_t)nd * bhip_assoc_coltop_size())); col = sc->d.p; }
	BHIP_TRY(bhip_assoc_phase1_ham(ctx, dev_src, ns, 0, dev_dst, nd, words, maxErr, dev_pairs, dev_fit, col, sc->work));
	if (col) BHIP_TRY(bhip_assoc_phase2(ctx, col, 1, nd, ns, 0, dev_pairs, dev_fit));
	return BHIP_OK;
}

int bhip_assoc_l2_dev_batched(bhip_ctx* ctx, const double* dev_src, const double* dev_dst, int dof, int count, const long long* srcOff, const int* ns,
							  const long long* dstOff, const int* nd, double maxErr, int backwards, int* dev_pairs, double* dev_fit) {
	CHECK_CTX(ctx);
	if (count < 0 || dof <= 0) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad sizes");
	if (count == 0) return BHIP_OK;
	if (!srcOff || !ns || !dstOff || !nd) return bhip_fail(ctx, BHIP_ERR_INVALID, "null problem table");
	if (dof == 64 && !assocExactOnly(ctx)) {
		int used = 0;
		BHIP_TRY(bhip_assoc_l2_mfma_batched(ctx, scratchOf(ctx)->mfma, dev_src, dev_dst, count, srcOff, ns, dstOff, nd, maxErr, backwards, dev_pairs,
											dev_fit, &used));
		if (used) return BHIP_OK;
	}
	for (int p = 0; p < count; p++) {
		if (ns[p] < 0 || nd[p] < 0) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad sizes");
		if (ns[p] == 0) continue;
		BHIP_TRY(assocL2Exact(ctx, dev_src + srcOff[p] * dof, ns[p], dev_dst + dstOff[p] * dof, nd[p], dof, maxErr, backwards, 0, dev_pairs + srcOff[p],
							  dev_fit + srcOff[p]));
	}
	return BHIP_OK;
}


// AssociateDescription over descriptors that are still resident from the last detect of `s` (FastQueue<BrightFeature> lists a provider
// recognises as its own): problem p associates image srcImage[p] (source) with image dstImage[p] (destination) -- same rules as
// bhip_assoc_l2_f64, no descriptor upload.  pairs / fit are host arrays over the compact key-point index space of the batch: the results
// of problem p start at the exclusive prefix of the counts of srcImage[p] (every image may be a source at most once per call).
// batched device form of bhip_assoc_hamming_dev (contract of bhip_assoc_l2_dev_batched): many small problems in three launches
int bhip_assoc_hamming_dev_batched(bhip_ctx* ctx, const int32_t* dev_src, const int32_t* dev_dst, int words, int count, const long long* srcOff, const int* ns,
								   const long long* dstOff, const int* nd, double maxErr, int backwards, int* dev_pairs, double* dev_fit) {
	CHECK_CTX(ctx);
	if (count < 0 || words <= 0) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad sizes");
	if (count == 0) return BHIP_OK;
	if (!srcOff || !ns || !dstOff || !nd || !dev_pairs || !dev_fit) return bhip_fail(ctx, BHIP_ERR_INVALID, "null problem table");
	return bhip_assoc_hamming_batched(ctx, dev_src, dev_dst, words, count, srcOff, ns, dstOff, nd, maxErr, backwards, dev_pairs, dev_fit, scratchOf(ctx)->work);
}

// AssociateDescription<TupleDesc_B>.associate() with ScoreAssociateHamming_B on the words still resident from the last detect of a BRIEF
// object: same contract as bhip_assoc_l2_surf, same rules and results as bhip_assoc_hamming, no descriptor upload.
int bhip_assoc_hamming_surf(bhip_surf* s, int count, const int* srcImage, const int* dstImage, double maxErr, int backwards, int* pairs, double* fit) {
	if (!s) return BHIP_ERR_INVALID;
	bhip_ctx* ctx = s->ctx;
	CHECK_CTX(ctx);
	if (!s->brief) return bhip_fail(ctx, BHIP_ERR_INVALID, "not a BRIEF detect+describe object");
	if (!s->haveResult) return bhip_fail(ctx, BHIP_ERR_INVALID, "no detect result");
	if (count < 0 || (count > 0 && (!srcImage || !dstImage || !pairs || !fit))) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad problem table");
	if (count == 0) return BHIP_OK;
	const long long total = s->det.total;
	std::vector<char> used(s->batch, 0);
	for (int p = 0; p < count; p++) {
		const int a = srcImage[p], b = dstImage[p];
		if (a < 0 || a >= s->batch || b < 0 || b >= s->batch) return bhip_fail(ctx, BHIP_ERR_INVALID, "image index outside the last batch");
		if (used[a]) return bhip_fail(ctx, BHIP_ERR_INVALID, "an image may be the source of one problem per call");
		used[a] = 1;
	}
	if (total == 0) return BHIP_OK;
	CtxScratch* sc = scratchOf(ctx);
	BHIP_TRY(sc->c.reserve(ctx, (size_t)total * 4));
	BHIP_TRY(sc->e.reserve(ctx, (size_t)total * 8));
	BHIP_HIP(ctx, hipMemsetAsync(sc->c.p, 0xff, (size_t)total * 4, ctx->stream));
	BHIP_HIP(ctx, hipMemsetAsync(sc->e.p, 0, (size_t)total * 8, ctx->stream));
	const int32_t* W = s->wordsBuf.as<int32_t>();
	std::vector<long long> so(count), doff(count);
	std::vector<int> ns(count), nd(count);
	for (int p = 0; p < count; p++) {
		const int a = srcImage[p], b = dstImage[p];
		so[p] = s->starts[a]; ns[p] = s->det.counts[a];
		doff[p] = s->starts[b]; nd[p] = s->det.counts[b];
	}
	BHIP_TRY(bhip_assoc_hamming_batched(ctx, W, W, s->briefWords, count, so.data(), ns.data(), doff.data(), nd.data(), maxErr, backwards, sc->c.as<int>(),
										sc->e.as<double>(), sc->work));
	BHIP_HIP(ctx, hipMemcpyAsync(pairs, sc->c.p, (size_t)total * 4, hipMemcpyDeviceToHost, ctx->stream));
	BHIP_HIP(ctx, hipMemcpyAsync(fit, sc->e.p, (size_t)total * 8, hipMemcpyDeviceToHost, ctx->stream));
	BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return BHIP_OK;
}

int bhip_assoc_l2_surf(bhip_surf* s, int count, const int* srcImage, const int* dstImage, double maxErr, int backwards, int* pairs, double* fit) {
	if (!s) return BHIP_ERR_INVALID;
	bhip_ctx* ctx = s->ctx;
	CHECK_CTX(ctx);
	if (s->brief) return bhip_fail(ctx, BHIP_ERR_INVALID, "this object describes with BRIEF: use bhip_assoc_hamming_surf");
	if (!s->haveResult) return bhip_fail(ctx, BHIP_ERR_INVALID, "no detect result");
	if (count < 0 || (count > 0 && (!srcImage || !dstImage || !pairs || !fit))) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad problem table");
	if (count == 0) return BHIP_OK;
	const long long total = s->det.total;
	std::vector<long long> so(count), doff(count);
	std::vector<int> ns(count), nd(count);
	std::vector<char> used(s->batch, 0);
	for (int p = 0; p < count; p++) {
		const int a = srcImage[p], b = dstImage[p];
		if (a < 0 || a >= s->batch || b < 0 || b >= s->batch) return bhip_fail(ctx, BHIP_ERR_INVALID, "image index outside the last batch");
		if (used[a]) return bhip_fail(ctx, BHIP_ERR_INVALID, "an image may be the source of one problem per call");
		used[a] = 1;
		so[p] = s->starts[a]; ns[p] = s->det.counts[a];
		doff[p] = s->starts[b]; nd[p] = s->det.counts[b];
	}
	if (total == 0) return BHIP_OK;
	CtxScratch* sc = scratchOf(ctx);
	BHIP_TRY(sc->c.reserve(ctx, (size_t)total * 4));
	BHIP_TRY(sc->e.reserve(ctx, (size_t)total * 8));
	// rows of images that are the source of no problem in this call come back as "no match" (-1, 0), not as whatever an earlier call left
	BHIP_HIP(ctx, hipMemsetAsync(sc->c.p, 0xff, (size_t)total * 4, ctx->stream));
	BHIP_HIP(ctx, hipMemsetAsync(sc->e.p, 0, (size_t)total * 8, ctx->stream));
	BHIP_TRY(bhip_assoc_l2_dev_batched(ctx, s->descBuf.as<double>(), s->descBuf.as<double>(), s->dofOut(), count, so.data(), ns.data(), doff.data(), nd.data(),
									   maxErr, backwards, sc->c.as<int>(), sc->e.as<double>()));
	BHIP_HIP(ctx, hipMemcpyAsync(pairs, sc->c.p, (size_t)total * 4, hipMemcpyDeviceToHost, ctx->stream));
	BHIP_HIP(ctx, hipMemcpyAsync(fit, sc->e.p, (size_t)total * 8, hipMemcpyDeviceToHost, ctx->stream));
	BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return BHIP_OK;
}

}  // extern "C"

template <class E>
static int assocHost(bhip_ctx* ctx, bool hamming, const E* src, int ns, const E* dst, int nd, int len, double maxErr, int backwards, int sqrtScore,
					 int* pairs, double* fit) {
	if (ns < 0 || nd < 0 || len <= 0) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad sizes");
	if (ns == 0) return BHIP_OK;
	if (!src || (nd > 0 && !dst) || !pairs || !fit) return bhip_fail(ctx, BHIP_ERR_INVALID, "null buffer");
	CtxScratch* sc = scratchOf(ctx);
	BHIP_TRY(sc->a.reserve(ctx, (size_t)ns * len * sizeof(E)));
	BHIP_TRY(sc->b.reserve(ctx, (size_t)std::max(nd, 1) * len * sizeof(E)));
	BHIP_TRY(sc->c.reserve(ctx, (size_t)ns * 4));
	BHIP_TRY(sc->e.reserve(ctx, (size_t)ns * 8));
	BHIP_HIP(ctx, hipMemcpyAsync(sc->a.p, src, (size_t)ns * len * sizeof(E), hipMemcpyHostToDevice, ctx->stream));
	if (nd > 0) BHIP_HIP(ctx, hipMemcpyAsync(sc->b.p, dst, (size_t)nd * len * sizeof(E), hipMemcpyHostToDevice, ctx->stream));
	if (hamming) BHIP_TRY(bhip_assoc_hamming_dev(ctx, (const int32_t*)sc->a.p, ns, (const int32_t*)sc->b.p, nd, len, maxErr, backwards, sc->c.as<int>(), sc->e.as<double>()));
	else BHIP_TRY(bhip_assoc_l2_dev(ctx, (const double*)sc->a.p, ns, (const double*)sc->b.p, nd, len, maxErr, backwards, sqrtScore, sc->c.as<int>(), sc->e.as<double>()));
	BHIP_HIP(ctx, hipMemcpyAsync(pairs, sc->c.p, (size_t)ns * 4, hipMemcpyDeviceToHost, ctx->stream));
	BHIP_HIP(ctx, hipMemcpyAsync(fit, sc->e.p, (size_t)ns * 8, hipMemcpyDeviceToHost, ctx->stream));
	BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return BHIP_OK;
}

extern "C" {

int bhip_assoc_l2_f64(bhip_ctx* ctx, const double* src, int ns, const double* dst, int nd, int dof, double maxErr, int backwards, int sqrtScore,
					  int* pairs, double* fit) {
	CHECK_CTX(ctx);
	return assocHost<double>(ctx, false, src, ns, dst, nd, dof, maxErr, backwards, sqrtScore, pairs, fit);
}
int bhip_assoc_hamming(bhip_ctx* ctx, const int32_t* src, int ns, const int32_t* dst, int nd, int words, double maxErr, int backwards, int* pairs,
					   double* fit) {
	CHECK_CTX(ctx);
	return assocHost<int32_t>(ctx, true, src, ns, dst, nd, words, maxErr, backwards, 0, pairs, fit);
}

int bhip_assoc_l2_shard_phase1(bhip_ctx* ctx, const double* dev_src, int nsLocal, int srcBegin, const double* dev_dst, int nd, int dof, double maxErr,
							   int* dev_pairs, double* dev_fit, void* dev_colTop) {
	CHECK_CTX(ctx);
	if (!dev_colTop) return bhip_fail(ctx, BHIP_ERR_INVALID, "null column buffer");
	return bhip_assoc_phase1_l2(ctx, dev_src, nsLocal, srcBegin, dev_dst, nd, dof, maxErr, 0, dev_pairs, dev_fit, dev_colTop, scratchOf(ctx)->work);
}
int bhip_assoc_hamming_shard_phase1(bhip_ctx* ctx, const int32_t* dev_src, int nsLocal, int srcBegin, const int32_t* dev_dst, int nd, int words,
									double maxErr, int* dev_pairs, double* dev_fit, void* dev_colTop) {
	CHECK_CTX(ctx);
	if (!dev_colTop) return bhip_fail(ctx, BHIP_ERR_INVALID, "null column buffer");
	return bhip_assoc_phase1_ham(ctx, dev_src, nsLocal, srcBegin, dev_dst, nd, words, maxErr, dev_pairs, dev_fit, dev_colTop, scratchOf(ctx)->work);
}
int bhip_assoc_shard_phase2(bhip_ctx* ctx, const void* dev_colTopAll, int nranks, int nd, int nsLocal, int srcBegin, int* dev_pairs, double* dev_fit) {
	CHECK_CTX(ctx);
	if (nranks < 1) return bhip_fail(ctx, BHIP_ERR_INVALID, "nranks < 1");
	return bhip_assoc_phase2(ctx, dev_colTopAll, nranks, nd, nsLocal, srcBegin, dev_pairs, dev_fit);
}

// ---------------------------------------------------------------------------------------------------------------
// boofcv-ip front end
// ---------------------------------------------------------------------------------------------------------------
static int convHost(bhip_ctx* ctx, bool vertical, bool normalized, const float* kernel, int kw, int koff, const float* in, int inStart, int inStride,
					int width, int height, float* out, int outStart, int outStride) {
	CHECK_IMG(ctx, in, inStride, width, height);
	CHECK_IMG(ctx, out, outStride, width, height);
	if (!kernel) return bhip_fail(ctx, BHIP_ERR_INVALID, "null kernel");
	CtxScratch* sc = scratchOf(ctx);
	const int pitch = pitch4(width);
	BHIP_TRY(uploadImage(ctx, sc->a, in, inStart, inStride, width, height, pitch));
	// the no-border variants leave the frame of `out` untouched: start from the caller's pixels
	BHIP_TRY(uploadImage(ctx, sc->b, out, outStart, outStride, width, height, pitch));
	BHIP_TRY(bhip_launch_conv(ctx, vertical, normalized, kernel, kw, koff, sc->a.as<float>(), pitch, width, height, sc->b.as<float>(), pitch));
	return downloadImage(ctx, sc->b.p, out, outStart, outStride, width, height, pitch);
}
int bhip_conv_h_f32(bhip_ctx* ctx, const float* kernel, int kw, int koff, const float* in, int inStart, int inStride, int width, int height, float* out,
					int outStart, int outStride) {
	CHECK_CTX(ctx);
	return convHost(ctx, false, false, kernel, kw, koff, in, inStart, inStride, width, height, out, outStart, outStride);
}
int bhip_conv_v_f32(bhip_ctx* ctx, const float* kernel, int kw, int koff, const float* in, int inStart, int inStride, int width, int height, float* out,
					int outStart, int outStride) {
	CHECK_CTX(ctx);
	return convHost(ctx, true, false, kernel, kw, koff, in, inStart, inStride, width, height, out, outStart, outStride);
}
int bhip_conv_norm_h_f32(bhip_ctx* ctx, const float* kernel, int kw, int koff, const float* in, int inStart, int inStride, int width, int height,
						 float* out, int outStart, int outStride) {
	CHECK_CTX(ctx);
	return convHost(ctx, false, true, kernel, kw, koff, in, inStart, inStride, width, height, out, outStart, outStride);
}
int bhip_conv_norm_v_f32(bhip_ctx* ctx, const float* kernel, int kw, int koff, const float* in, int inStart, int inStride, int width, int height,
						 float* out, int outStart, int outStride) {
	CHECK_CTX(ctx);
	return convHost(ctx, true, true, kernel, kw, koff, in, inStart, inStride, width, height, out, outStart, outStride);
}

int bhip_gaussian_f32(bhip_ctx* ctx, const float* in, int inStart, int inStride, int width, int height, double sigma, int radius, float* out,
					  int outStart, int outStride) {
	CHECK_CTX(ctx);
	CHECK_IMG(ctx, in, inStride, width, height);
	CHECK_IMG(ctx, out, outStride, width, height);
	if (sigma <= 0 && radius <= 0) return bhip_fail(ctx, BHIP_ERR_INVALID, "Sigma must be > 0");
	std::vector<float> k = bhip_gaussian1d_f32(sigma, radius);
	const int kw = (int)k.size(), koff = kw / 2;
	CtxScratch* sc = scratchOf(ctx);
	const int pitch = pitch4(width);
	BHIP_TRY(uploadImage(ctx, sc->a, in, inStart, inStride, width, height, pitch));
	BHIP_TRY(sc->c.reserve(ctx, (size_t)pitch * height * 4));
	bool fused = false;
	BHIP_TRY(bhip_launch_blur_fused(ctx, k.data(), kw, sc->a.as<float>(), pitch, width, height, sc->c.as<float>(), pitch, 1, 0, 0, &fused));
	if (!fused) {
		BHIP_TRY(sc->b.reserve(ctx, (size_t)pitch * height * 4));
		BHIP_TRY(bhip_launch_conv(ctx, false, true, k.data(), kw, koff, sc->a.as<float>(), pitch, width, height, sc->b.as<float>(), pitch));
		BHIP_TRY(bhip_launch_conv(ctx, true, true, k.data(), kw, koff, sc->b.as<float>(), pitch, width, height, sc->c.as<float>(), pitch));
	}
	return downloadImage(ctx, sc->c.p, out, outStart, outStride, width, height, pitch);
}

static int convDownHost(bhip_ctx* ctx, bool vertical, const float* kernel, int kw, const float* in, int inStart, int inStride, int width, int height,
						float* out, int outStart, int outStride, int outWidth, int outHeight, int skip) {
	CHECK_IMG(ctx, in, inStride, width, height);
	CHECK_IMG(ctx, out, outStride, outWidth, outHeight);
	if (!kernel) return bhip_fail(ctx, BHIP_ERR_INVALID, "null kernel");
	CtxScratch* sc = scratchOf(ctx);
	BHIP_TRY(uploadImage(ctx, sc->a, in, inStart, inStride, width, height));
	// pixels the reference does not write keep the caller's values
	BHIP_TRY(uploadImage(ctx, sc->b, out, outStart, outStride, outWidth, outHeight));
	BHIP_TRY(bhip_launch_conv_down(ctx, vertical, kernel, kw, sc->a.as<float>(), 0, width, width, height, sc->b.as<float>(), 0, outWidth, outWidth,
								   outHeight, skip, 1));
	return downloadImage(ctx, sc->b.p, out, outStart, outStride, outWidth, outHeight);
}
int bhip_conv_down_norm_h_f32(bhip_ctx* ctx, const float* kernel, int kw, const float* in, int inStart, int inStride, int width, int height, float* out,
							  int outStart, int outStride, int outWidth, int outHeight, int skip) {
	CHECK_CTX(ctx);
	return convDownHost(ctx, false, kernel, kw, in, inStart, inStride, width, height, out, outStart, outStride, outWidth, outHeight, skip);
}
int bhip_conv_down_norm_v_f32(bhip_ctx* ctx, const float* kernel, int kw, const float* in, int inStart, int inStride, int width, int height, float* out,
							  int outStart, int outStride, int outWidth, int outHeight, int skip) {
	CHECK_CTX(ctx);
	return convDownHost(ctx, true, kernel, kw, in, inStart, inStride, width, height, out, outStart, outStride, outWidth, outHeight, skip);
}

// FactoryKernelGaussian.gaussian(Kernel1D_F32.class, sigma, radius) (I:factory/filter/kernel/FactoryKernelGaussian.java:120-153)
int bhip_gaussian_kernel1d_f32(double sigma, int radius, float* out, int capacity) {
	if (sigma <= 0 && radius <= 0) return -1;
	std::vector<float> k = bhip_gaussian1d_f32(sigma, radius);
	if (!out || (int)k.size() > capacity) return -(int)k.size();
	for (size_t i = 0; i < k.size(); i++) out[i] = k[i];
	return (int)k.size();
}

// PyramidDiscreteSampleBlur: layer geometry (ImagePyramidBase.initialize) + scale checks (ImagePyramidBase.checkScales)
int bhip_pyramid_layout(int width, int height, const int* scales, int n, int* dims, long long* offsets, long long* totalFloats) {
	if (width <= 0 || height <= 0 || !scales || n <= 0) return BHIP_ERR_INVALID;
	if (scales[0] <= 0) return BHIP_ERR_INVALID;
	int prev = 0;
	long long off = 0;
	for (int i = 0; i < n; i++) {
		if (scales[i] < prev) return BHIP_ERR_INVALID;
		prev = scales[i];
		const double sf = scales[i];
		int w = (int)std::ceil(width / sf), h = (int)std::ceil(height / sf);
		if (i == 0 && scales[0] == 1) { w = width; h = height; }
		if (dims) { dims[2 * i] = w; dims[2 * i + 1] = h; }
		if (offsets) offsets[i] = off;
		off += (long long)w * h;
	}
	if (totalFloats) *totalFloats = off;
	return BHIP_OK;
}

int bhip_pyramid_dev_f32(bhip_ctx* ctx, const float* kernel, int kw, const int* scales, int n, const float* dev_in, long long inImageStride, int inStride,
						 int width, int height, int batch, float* dev_out) {
	CHECK_CTX(ctx);
	if (!kernel || !dev_in || !dev_out || batch <= 0 || inStride < width) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad pyramid arguments");
	if (n > 32) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "more than 32 layers");
	int dims[64];
	long long offs[32], total = 0;
	if (bhip_pyramid_layout(width, height, scales, n, dims, offs, &total) != BHIP_OK) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad pyramid scales");
	for (int i = 1; i < n; i++)
		if (scales[i] / scales[i - 1] <= 0) return bhip_fail(ctx, BHIP_ERR_INVALID, "Skip must be >= 1");
	CtxScratch* sc = scratchOf(ctx);
	// freshly created layers are zero (pixels outside floor(prev/skip) are never written)
	BHIP_HIP(ctx, hipMemsetAsync(dev_out, 0, (size_t)total * batch * 4, ctx->stream));
	// `temp` of the reference is one grow-only image shared by all layers: zero when first allocated, afterwards it keeps the
	// previous layer's values wherever the off-grid skip>=3 case leaves a column unwritten.  Same here: one dense region per
	// frame, sized for the first convolved layer, cleared once per call (= first process() of a fresh pyramid object).
	long long tempCap = 0;
	{
		int pw0 = width, ph0 = height;
		for (int i = 0; i < n; i++) {
			if (!(i == 0 && scales[0] == 1)) {
				const int skip = i == 0 ? scales[0] : scales[i] / scales[i - 1];
				tempCap = std::max(tempCap, (long long)(pw0 / skip) * ph0);
			}
			pw0 = dims[2 * i]; ph0 = dims[2 * i + 1];
		}
	}
	if (tempCap > 0) {
		BHIP_TRY(sc->d.reserve(ctx, (size_t)tempCap * 4 * batch));
		BHIP_HIP(ctx, hipMemsetAsync(sc->d.p, 0, (size_t)tempCap * 4 * batch, ctx->stream));
	}
	const float* prev = dev_in;
	long long prevImageStride = inImageStride;
	int prevStride = inStride, pw = width, ph = height;
	for (int i = 0; i < n; i++) {
		float* layer = dev_out + offs[i];
		const int lw = dims[2 * i], lh = dims[2 * i + 1];
		if (i == 0 && scales[0] == 1) {
			ProfScope prof(ctx, "pyramid_copy", 8.0 * width * height * batch);
			BHIP_TRY(bhip_launch_copy_images(ctx, dev_in, inImageStride, inStride, layer, total, lw, width, height, batch));
		} else {
			const int skip = i == 0 ? scales[0] : scales[i] / scales[i - 1];
			const int tw = pw / skip;   // 0 when the layer below is narrower than the step: both passes then write nothing and the (ceil-sized) layer stays zero, as in the reference
			bool fused = false;
			BHIP_TRY(bhip_launch_pyr_layer_fused(ctx, kernel, kw, prev, prevImageStride, prevStride, pw, ph, layer, total, lw, skip, batch, &fused));
			if (!fused) {
				BHIP_TRY(bhip_launch_conv_down(ctx, false, kernel, kw, prev, prevImageStride, prevStride, pw, ph, sc->d.as<float>(), tempCap, tw, tw, ph, skip,
											   batch));
				BHIP_TRY(bhip_launch_conv_down(ctx, true, kernel, kw, sc->d.as<float>(), tempCap, tw, tw, ph, layer, total, lw, lw, lh, skip, batch));
			}
		}
		prev = layer; prevImageStride = total; prevStride = lw; pw = lw; ph = lh;
	}
	return BHIP_OK;
}

int bhip_pyramid_f32(bhip_ctx* ctx, const float* kernel, int kw, const int* scales, int n, const float* in, int inStart, int inStride, int width,
					 int height, float* out) {
	CHECK_CTX(ctx);
	CHECK_IMG(ctx, in, inStride, width, height);
	if (!out || !kernel || !scales) return bhip_fail(ctx, BHIP_ERR_INVALID, "null buffer");
	long long total = 0;
	if (bhip_pyramid_layout(width, height, scales, n, nullptr, nullptr, &total) != BHIP_OK) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad pyramid scales");
	CtxScratch* sc = scratchOf(ctx);
	BHIP_TRY(uploadImage(ctx, sc->a, in, inStart, inStride, width, height));
	BHIP_TRY(sc->b.reserve(ctx, (size_t)total * 4));
	BHIP_TRY(bhip_pyramid_dev_f32(ctx, kernel, kw, scales, n, sc->a.as<float>(), (long long)width * height, width, width, height, 1, sc->b.as<float>()));
	BHIP_HIP(ctx, hipMemcpyAsync(out, sc->b.p, (size_t)total * 4, hipMemcpyDeviceToHost, ctx->stream));
	BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return BHIP_OK;
}

int bhip_conv2d_f32(bhip_ctx* ctx, const float* kernel, int kw, int koff, const float* in, int inStart, int inStride, int width, int height, float* out,
					int outStart, int outStride) {
	CHECK_CTX(ctx);
	CHECK_IMG(ctx, in, inStride, width, height);
	CHECK_IMG(ctx, out, outStride, width, height);
	if (!kernel) return bhip_fail(ctx, BHIP_ERR_INVALID, "null kernel");
	CtxScratch* sc = scratchOf(ctx);
	BHIP_TRY(uploadImage(ctx, sc->a, in, inStart, inStride, width, height));
	BHIP_TRY(uploadImage(ctx, sc->b, out, outStart, outStride, width, height));   // the frame keeps the caller's pixels
	BHIP_TRY(bhip_launch_conv2d(ctx, kernel, kw, koff, sc->a.as<float>(), width, width, height, sc->b.as<float>(), width));
	return downloadImage(ctx, sc->b.p, out, outStart, outStride, width, height);
}
int bhip_mean_f32(bhip_ctx* ctx, const float* in, int inStart, int inStride, int width, int height, int radiusX, int radiusY, float* out, int outStart,
				  int outStride) {
	CHECK_CTX(ctx);
	CHECK_IMG(ctx, in, inStride, width, height);
	CHECK_IMG(ctx, out, outStride, width, height);
	if (radiusX <= 0 || radiusY <= 0) return bhip_fail(ctx, BHIP_ERR_INVALID, "Radius must be > 0");
	CtxScratch* sc = scratchOf(ctx);
	BHIP_TRY(uploadImage(ctx, sc->a, in, inStart, inStride, width, height));
	BHIP_TRY(sc->b.reserve(ctx, (size_t)width * height * 4));
	BHIP_TRY(sc->c.reserve(ctx, (size_t)width * height * 4));
	BHIP_TRY(bhip_launch_mean(ctx, false, sc->a.as<float>(), sc->b.as<float>(), width, height, radiusX));
	BHIP_TRY(bhip_launch_mean(ctx, true, sc->b.as<float>(), sc->c.as<float>(), width, height, radiusY));
	return downloadImage(ctx, sc->c.p, out, outStart, outStride, width, height);
}
int bhip_median_f32(bhip_ctx* ctx, const float* in, int inStart, int inStride, int width, int height, int radius, float* out, int outStart, int outStride) {
	CHECK_CTX(ctx);
	CHECK_IMG(ctx, in, inStride, width, height);
	CHECK_IMG(ctx, out, outStride, width, height);
	if (radius <= 0) return bhip_fail(ctx, BHIP_ERR_INVALID, "Radius must be > 0");
	CtxScratch* sc = scratchOf(ctx);
	BHIP_TRY(uploadImage(ctx, sc->a, in, inStart, inStride, width, height));
	BHIP_TRY(sc->b.reserve(ctx, (size_t)width * height * 4));
	BHIP_TRY(bhip_launch_median(ctx, sc->a.as<float>(), width, sc->b.as<float>(), width, width, height, radius));
	return downloadImage(ctx, sc->b.p, out, outStart, outStride, width, height);
}

static int gradHost(bhip_ctx* ctx, int kind, const float* in, int inStart, int inStride, int width, int height, float* dx, float* dy, int outStart,
					int outStride, int border) {
	CHECK_IMG(ctx, in, inStride, width, height);
	CHECK_IMG(ctx, dx, outStride, width, height);
	CHECK_IMG(ctx, dy, outStride, width, height);
	if (border != 0 && border != 1) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "border policy not supported on the GPU");
	CtxScratch* sc = scratchOf(ctx);
	const int pitch = pitch4(width);
	BHIP_TRY(uploadImage(ctx, sc->a, in, inStart, inStride, width, height, pitch));
	BHIP_TRY(uploadImage(ctx, sc->b, dx, outStart, outStride, width, height, pitch));
	BHIP_TRY(uploadImage(ctx, sc->c, dy, outStart, outStride, width, height, pitch));
	BHIP_TRY(bhip_launch_gradient(ctx, kind, sc->a.as<float>(), pitch, width, height, sc->b.as<float>(), sc->c.as<float>(), pitch, border));
	BHIP_TRY(downloadImage(ctx, sc->b.p, dx, outStart, outStride, width, height, pitch));
	return downloadImage(ctx, sc->c.p, dy, outStart, outStride, width, height, pitch);
}
int bhip_sobel_f32(bhip_ctx* ctx, const float* in, int inStart, int inStride, int width, int height, float* dx, float* dy, int outStart, int outStride,
				   int border) {
	CHECK_CTX(ctx);
	return gradHost(ctx, 0, in, inStart, inStride, width, height, dx, dy, outStart, outStride, border);
}
int bhip_three_f32(bhip_ctx* ctx, const float* in, int inStart, int inStride, int width, int height, float* dx, float* dy, int outStart, int outStride,
				   int border) {
	CHECK_CTX(ctx);
	return gradHost(ctx, 1, in, inStart, inStride, width, height, dx, dy, outStart, outStride, border);
}

// FactoryIntensityPointAlg.shiTomasi / harris (unweighted, GrayF32) -> GradientCornerIntensity.process(derivX, derivY, intensity)
int bhip_corner_intensity_f32(bhip_ctx* ctx, int kind, int radius, float kappa, const float* derivX, const float* derivY, int dStart, int dStride, int width,
							  int height, float* intensity, int iStart, int iStride) {
	CHECK_CTX(ctx);
	CHECK_IMG(ctx, derivX, dStride, width, height);
	CHECK_IMG(ctx, derivY, dStride, width, height);
	CHECK_IMG(ctx, intensity, iStride, width, height);
	CtxScratch* sc = scratchOf(ctx);
	const size_t px = (size_t)width * height;
	BHIP_TRY(uploadImage(ctx, sc->a, derivX, dStart, dStride, width, height));
	BHIP_TRY(uploadImage(ctx, sc->b, derivY, dStart, dStride, width, height));
	BHIP_TRY(sc->c.reserve(ctx, px * 4 * 3));
	BHIP_TRY(sc->d.reserve(ctx, px * 4));
	BHIP_HIP(ctx, hipMemsetAsync(sc->d.p, 0, px * 4, ctx->stream));   // ImageMiscOps.fillBorder(intensity, 0, radius); the interior is overwritten
	float* h = sc->c.as<float>();
	BHIP_TRY(bhip_launch_corner_intensity(ctx, kind, radius, kappa, sc->a.as<float>(), sc->b.as<float>(), width, width, height, h, h + px, h + 2 * px,
										  sc->d.as<float>(), width));
	return downloadImage(ctx, sc->d.p, intensity, iStart, iStride, width, height);
}

// ---- integer image variants, stage level ----
int bhip_integral_u8_s32(bhip_ctx* ctx, const uint8_t* in, int inStart, int inStride, int width, int height, int32_t* out, int outStart, int outStride) {
	CHECK_CTX(ctx);
	if (!in || !out || width <= 0 || height <= 0 || inStride < width || outStride < width) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad image");
	CtxScratch* sc = scratchOf(ctx);
	BHIP_TRY(sc->a.reserve(ctx, (size_t)width * height));
	BHIP_TRY(sc->b.reserve(ctx, (size_t)width * height * 4));
	BHIP_HIP(ctx, hipMemcpy2DAsync(sc->a.p, (size_t)width, in + inStart, (size_t)inStride, (size_t)width, height, hipMemcpyHostToDevice, ctx->stream));
	BHIP_TRY(bhip_launch_integral_u8(ctx, (const unsigned char*)sc->a.p, 0, width, sc->b.as<int>(), 0, width, width, height, 1));
	BHIP_HIP(ctx, hipMemcpy2DAsync(out + outStart, (size_t)outStride * 4, sc->b.p, (size_t)width * 4, (size_t)width * 4, height, hipMemcpyDeviceToHost, ctx->stream));
	BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return BHIP_OK;
}
int bhip_hessian_s32(bhip_ctx* ctx, const int32_t* ii, int iiStart, int iiStride, int width, int height, int skip, int size, float* out, int outStart,
					 int outStride) {
	CHECK_CTX(ctx);
	if (!ii || !out || width <= 0 || height <= 0 || iiStride < width) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad image");
	if (skip < 1 || size < 3) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad skip / size");
	const int w = width / skip, h = height / skip;
	if (w <= 0 || h <= 0 || outStride < w) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad intensity image");
	CtxScratch* sc = scratchOf(ctx);
	BHIP_TRY(uploadImage(ctx, sc->a, (const float*)ii, iiStart, iiStride, width, height));   // 32-bit words either way
	BHIP_TRY(sc->b.reserve(ctx, (size_t)w * h * 4));
	ImgView iv{sc->a.as<float>(), (long long)width * height, width, width, height};
	BHIP_TRY(bhip_launch_hessian(ctx, iv, 1, skip, 1, &size, sc->b.as<float>(), (long long)w * h, (long long)w * h, w, nullptr, true));
	return downloadImage(ctx, sc->b.p, out, outStart, outStride, w, h);
}
int bhip_brief_u8(bhip_ctx* ctx, const uint8_t* img, int start, int stride, int width, int height, int radius, int numPoints, const int32_t* samplePoints,
				  const int32_t* compare, const double* xy, int n, int32_t* out) {
	CHECK_CTX(ctx);
	if (!img || width <= 0 || height <= 0 || stride < width) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad image");
	if (numPoints <= 0 || !samplePoints || !compare || n < 0 || (n > 0 && (!xy || !out))) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad BRIEF arguments");
	if (n == 0) return BHIP_OK;
	int maxIdx = 0;
	for (int i = 0; i < 2 * numPoints; i++) { if (compare[i] < 0) return bhip_fail(ctx, BHIP_ERR_INVALID, "negative sample index"); maxIdx = std::max(maxIdx, compare[i]); }
	const int words = (numPoints + 31) / 32;
	CtxScratch* sc = scratchOf(ctx);
	BHIP_TRY(sc->a.reserve(ctx, (size_t)width * height));
	BHIP_TRY(sc->b.reserve(ctx, (size_t)(maxIdx + 1) * 8 + (size_t)numPoints * 8));
	BHIP_TRY(sc->c.reserve(ctx, (size_t)n * 16));
	BHIP_TRY(sc->d.reserve(ctx, (size_t)n * words * 4));
	BHIP_HIP(ctx, hipMemcpy2DAsync(sc->a.p, (size_t)width, img + start, (size_t)stride, (size_t)width, height, hipMemcpyHostToDevice, ctx->stream));
	int* dSample = sc->b.as<int>();
	int* dCompare = dSample + 2 * (maxIdx + 1);
	BHIP_HIP(ctx, hipMemcpyAsync(dSample, samplePoints, (size_t)(maxIdx + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
	BHIP_HIP(ctx, hipMemcpyAsync(dCompare, compare, (size_t)numPoints * 8, hipMemcpyHostToDevice, ctx->stream));
	BHIP_HIP(ctx, hipMemcpyAsync(sc->c.p, xy, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
	BHIP_TRY(bhip_launch_brief(ctx, (const float*)sc->a.p, width, width, height, radius, numPoints, dSample, dCompare, sc->c.as<double>(), n, sc->d.as<int>(), true, 1, 0, nullptr, 0, 2, 0,
							   briefPatchOk(samplePoints, maxIdx + 1, radius)));
	BHIP_HIP(ctx, hipMemcpyAsync(out, sc->d.p, (size_t)n * words * 4, hipMemcpyDeviceToHost, ctx->stream));
	BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return BHIP_OK;
}

int bhip_brief_f32(bhip_ctx* ctx, const float* img, int start, int stride, int width, int height, int radius, int numPoints,
				   const int32_t* samplePoints, const int32_t* compare, const double* xy, int n, int32_t* out) {
	CHECK_CTX(ctx);
	CHECK_IMG(ctx, img, stride, width, height);
	if (numPoints <= 0 || !samplePoints || !compare || n < 0 || (n > 0 && (!xy || !out))) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad BRIEF arguments");
	if (n == 0) return BHIP_OK;
	for (int i = 0; i < 2 * numPoints; i++)
		if (compare[i] < 0 || compare[i] >= numPoints) return bhip_fail(ctx, BHIP_ERR_INVALID, "pair index outside the sample point list");
	const int words = (numPoints + 31) / 32;
	CtxScratch* sc = scratchOf(ctx);
	BHIP_TRY(uploadImage(ctx, sc->a, img, start, stride, width, height));
	BHIP_TRY(sc->b.reserve(ctx, (size_t)numPoints * 16));
	BHIP_TRY(sc->c.reserve(ctx, (size_t)n * 16));
	BHIP_TRY(sc->e.reserve(ctx, (size_t)n * words * 4));
	int* dsp = sc->b.as<int>();
	int* dcp = dsp + 2 * numPoints;
	BHIP_HIP(ctx, hipMemcpyAsync(dsp, samplePoints, (size_t)numPoints * 8, hipMemcpyHostToDevice, ctx->stream));
	BHIP_HIP(ctx, hipMemcpyAsync(dcp, compare, (size_t)numPoints * 8, hipMemcpyHostToDevice, ctx->stream));
	BHIP_HIP(ctx, hipMemcpyAsync(sc->c.p, xy, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
	BHIP_TRY(bhip_launch_brief(ctx, sc->a.as<float>(), width, width, height, radius, numPoints, dsp, dcp, sc->c.as<double>(), n, sc->e.as<int>(), false, 1, 0, nullptr, 0, 2, 0,
							   briefPatchOk(samplePoints, numPoints, radius)));
	BHIP_HIP(ctx, hipMemcpyAsync(out, sc->e.p, (size_t)n * words * 4, hipMemcpyDeviceToHost, ctx->stream));
	BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));
	return BHIP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// device-resident, batched forms of the boofcv-ip front end (BASELINE config 5: pyramid -> gradient -> NMS -> SURF on a 4K stream without
// leaving HBM).  Image b of a batch starts imageStride floats after image 0, rows are `stride` floats apart; everything is
// asynchronous on the ctx stream.  Same kernels and arithmetic as the host-buffer entry points above.
// ---------------------------------------------------------------------------------------------------------------
#define CHECK_DEV_BATCH(ctx, p, stride, w, h, batch)                                                                                      \
	do {                                                                                                                                  \
		if (!(p) || (w) <= 0 || (h) <= 0 || (batch) <= 0 || (stride) < (w)) return bhip_fail((ctx), BHIP_ERR_INVALID, "bad image batch"); \
	} while (0)

static int convDev(bhip_ctx* ctx, bool vertical, bool normalized, const float* kernel, int kw, int koff, const float* dev_in, long long inImageStride,
				   int inStride, int width, int height, int batch, float* dev_out, long long outImageStride, int outStride) {
	CHECK_DEV_BATCH(ctx, dev_in, inStride, width, height, batch);
	CHECK_DEV_BATCH(ctx, dev_out, outStride, width, height, batch);
	if (!kernel) return bhip_fail(ctx, BHIP_ERR_INVALID, "null kernel");
	return bhip_launch_conv(ctx, vertical, normalized, kernel, kw, koff, dev_in, inStride, width, height, dev_out, outStride, batch, inImageStride, outImageStride);
}
int bhip_conv_h_dev_f32(bhip_ctx* ctx, const float* kernel, int kw, int koff, const float* dev_in, long long inImageStride, int inStride, int width, int height,
						int batch, float* dev_out, long long outImageStride, int outStride) {
	CHECK_CTX(ctx);
	return convDev(ctx, false, false, kernel, kw, koff, dev_in, inImageStride, inStride, width, height, batch, dev_out, outImageStride, outStride);
}
int bhip_conv_v_dev_f32(bhip_ctx* ctx, const float* kernel, int kw, int koff, const float* dev_in, long long inImageStride, int inStride, int width, int height,
						int batch, float* dev_out, long long outImageStride, int outStride) {
	CHECK_CTX(ctx);
	return convDev(ctx, true, false, kernel, kw, koff, dev_in, inImageStride, inStride, width, height, batch, dev_out, outImageStride, outStride);
}
int bhip_conv_norm_h_dev_f32(bhip_ctx* ctx, const float* kernel, int kw, int koff, const float* dev_in, long long inImageStride, int inStride, int width,
							 int height, int batch, float* dev_out, long long outImageStride, int outStride) {
	CHECK_CTX(ctx);
	return convDev(ctx, false, true, kernel, kw, koff, dev_in, inImageStride, inStride, width, height, batch, dev_out, outImageStride, outStride);
}
int bhip_conv_norm_v_dev_f32(bhip_ctx* ctx, const float* kernel, int kw, int koff, const float* dev_in, long long inImageStride, int inStride, int width,
							 int height, int batch, float* dev_out, long long outImageStride, int outStride) {
	CHECK_CTX(ctx);
	return convDev(ctx, true, true, kernel, kw, koff, dev_in, inImageStride, inStride, width, height, batch, dev_out, outImageStride, outStride);
}

// BlurImageOps.gaussian on a device batch: horizontal pass into the library's `storage`, vertical pass into dev_out
int bhip_gaussian_dev_f32(bhip_ctx* ctx, const float* dev_in, long long inImageStride, int inStride, int width, int height, int batch, double sigma, int radius,
						  float* dev_out, long long outImageStride, int outStride) {
	CHECK_CTX(ctx);
	CHECK_DEV_BATCH(ctx, dev_in, inStride, width, height, batch);
	CHECK_DEV_BATCH(ctx, dev_out, outStride, width, height, batch);
	if (sigma <= 0 && radius <= 0) return bhip_fail(ctx, BHIP_ERR_INVALID, "Sigma must be > 0");
	std::vector<float> k = bhip_gaussian1d_f32(sigma, radius);
	const int kw = (int)k.size(), koff = kw / 2;
	CtxScratch* sc = scratchOf(ctx);
	bool fused = false;
	BHIP_TRY(bhip_launch_blur_fused(ctx, k.data(), kw, dev_in, inStride, width, height, dev_out, outStride, batch, inImageStride, outImageStride, &fused));
	if (fused) return BHIP_OK;
	const int pitch = pitch4(width);
	const long long tmpImage = (long long)pitch * height;
	BHIP_TRY(sc->ipTmp.reserve(ctx, (size_t)tmpImage * 4 * batch));
	BHIP_TRY(bhip_launch_conv(ctx, false, true, k.data(), kw, koff, dev_in, inStride, width, height, sc->ipTmp.as<float>(), pitch, batch, inImageStride, tmpImage));
	return bhip_launch_conv(ctx, true, true, k.data(), kw, koff, sc->ipTmp.as<float>(), pitch, width, height, dev_out, outStride, batch, tmpImage, outImageStride);
}

static int gradDev(bhip_ctx* ctx, int kind, const float* dev_in, long long inImageStride, int inStride, int width, int height, int batch, float* dev_dx,
				   float* dev_dy, long long outImageStride, int outStride, int border) {
	CHECK_DEV_BATCH(ctx, dev_in, inStride, width, height, batch);
	CHECK_DEV_BATCH(ctx, dev_dx, outStride, width, height, batch);
	CHECK_DEV_BATCH(ctx, dev_dy, outStride, width, height, batch);
	if (border != 0 && border != 1) return bhip_fail(ctx, BHIP_ERR_UNSUPPORTED, "border policy not supported on the GPU");
	return bhip_launch_gradient(ctx, kind, dev_in, inStride, width, height, dev_dx, dev_dy, outStride, border, batch, inImageStride, outImageStride);
}
int bhip_sobel_dev_f32(bhip_ctx* ctx, const float* dev_in, long long inImageStride, int inStride, int width, int height, int batch, float* dev_dx, float* dev_dy,
					   long long outImageStride, int outStride, int border) {
	CHECK_CTX(ctx);
	return gradDev(ctx, 0, dev_in, inImageStride, inStride, width, height, batch, dev_dx, dev_dy, outImageStride, outStride, border);
}
int bhip_three_dev_f32(bhip_ctx* ctx, const float* dev_in, long long inImageStride, int inStride, int width, int height, int batch, float* dev_dx, float* dev_dy,
					   long long outImageStride, int outStride, int border) {
	CHECK_CTX(ctx);
	return gradDev(ctx, 1, dev_in, inImageStride, inStride, width, height, batch, dev_dx, dev_dy, outImageStride, outStride, border);
}

int bhip_gradient_intensity_dev_f32(bhip_ctx* ctx, int kind, const float* dev_dx, const float* dev_dy, long long dImageStride, int dStride, int width, int height,
									int batch, float* dev_out, long long outImageStride, int outStride) {
	CHECK_CTX(ctx);
	CHECK_DEV_BATCH(ctx, dev_dx, dStride, width, height, batch);
	CHECK_DEV_BATCH(ctx, dev_dy, dStride, width, height, batch);
	CHECK_DEV_BATCH(ctx, dev_out, outStride, width, height, batch);
	return bhip_launch_grad_intensity(ctx, kind, dev_dx, dev_dy, dImageStride, dStride, dev_out, outImageStride, outStride, width, height, batch);
}

int bhip_corner_intensity_dev_f32(bhip_ctx* ctx, int kind, int radius, float kappa, const float* dev_dx, const float* dev_dy, long long dImageStride, int dStride,
								  int width, int height, int batch, float* dev_intensity, long long iImageStride, int iStride) {
	CHECK_CTX(ctx);
	CHECK_DEV_BATCH(ctx, dev_dx, dStride, width, height, batch);
	CHECK_DEV_BATCH(ctx, dev_dy, dStride, width, height, batch);
	CHECK_DEV_BATCH(ctx, dev_intensity, iStride, width, height, batch);
	CtxScratch* sc = scratchOf(ctx);
	const size_t px = (size_t)width * height;
	BHIP_TRY(sc->ipTmp.reserve(ctx, px * 4 * 3 * batch));
	// ImageMiscOps.fillBorder(intensity, 0, radius): clear every image, the interior is overwritten
	BHIP_HIP(ctx, hipMemset2DAsync(dev_intensity, (size_t)iStride * 4, 0, (size_t)width * 4, (size_t)height, ctx->stream));
	for (int b = 1; b < batch; b++)
		BHIP_HIP(ctx, hipMemset2DAsync(dev_intensity + (long long)b * iImageStride, (size_t)iStride * 4, 0, (size_t)width * 4, (size_t)height, ctx->stream));
	float* h = sc->ipTmp.as<float>();
	return bhip_launch_corner_intensity(ctx, kind, radius, kappa, dev_dx, dev_dy, dStride, width, height, h, h + px, h + 2 * px, dev_intensity, iStride, batch,
										dImageStride, (long long)px * 3, iImageStride);
}

// DescribePointBrief.process over a batch: the points of image b are dev_xy[start[b] .. start[b+1]) (host prefix `start`, batch+1 entries);
// words of point p at dev_out[p * ceil(numPoints/32)]
int bhip_brief_dev_f32(bhip_ctx* ctx, const float* dev_img, long long imageStride, int stride, int width, int height, int batch, int radius, int numPoints,
					   const int32_t* samplePoints, const int32_t* compare, const double* dev_xy, const int* start, int32_t* dev_out) {
	CHECK_CTX(ctx);
	CHECK_DEV_BATCH(ctx, dev_img, stride, width, height, batch);
	if (numPoints <= 0 || !samplePoints || !compare || !start) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad BRIEF arguments");
	int maxCount = 0;
	for (int b = 0; b < batch; b++) {
		if (start[b + 1] < start[b]) return bhip_fail(ctx, BHIP_ERR_INVALID, "point prefix must not decrease");
		maxCount = std::max(maxCount, start[b + 1] - start[b]);
	}
	const int n = start[batch] - start[0];
	if (n == 0) return BHIP_OK;
	if (!dev_xy || !dev_out) return bhip_fail(ctx, BHIP_ERR_INVALID, "bad BRIEF arguments");
	int maxIdx = 0;
	for (int i = 0; i < 2 * numPoints; i++) { if (compare[i] < 0) return bhip_fail(ctx, BHIP_ERR_INVALID, "negative sample index"); maxIdx = std::max(maxIdx, compare[i]); }
	CtxScratch* sc = scratchOf(ctx);
	const size_t nSample = (size_t)(maxIdx + 1) * 2, nCompare = (size_t)numPoints * 2;
	BHIP_TRY(sc->ipKernel.reserve(ctx, (nSample + nCompare + batch + 1) * 4));
	int* dSample = sc->ipKernel.as<int>();
	int* dCompare = dSample + nSample;
	int* dStart = dCompare + nCompare;
	BHIP_HIP(ctx, hipMemcpyAsync(dSample, samplePoints, nSample * 4, hipMemcpyHostToDevice, ctx->stream));
	BHIP_HIP(ctx, hipMemcpyAsync(dCompare, compare, nCompare * 4, hipMemcpyHostToDevice, ctx->stream));
	BHIP_HIP(ctx, hipMemcpyAsync(dStart, start, (size_t)(batch + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
	BHIP_TRY(bhip_launch_brief(ctx, dev_img, stride, width, height, radius, numPoints, dSample, dCompare, dev_xy, n, dev_out, false, batch, imageStride, dStart, maxCount, 2, 0,
							   briefPatchOk(samplePoints, maxIdx + 1, radius)));
	BHIP_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the host tables were handed to async copies
	return BHIP_OK;
}

}  // extern "C"
