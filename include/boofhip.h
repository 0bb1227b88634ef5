/*
 * boofhip.h -- C ABI of libboofhip.so, the MI355X (gfx950) provider for BoofCV's
 * detect -> describe -> associate hot path.
 *
 * This is the drop-in boundary: every entry point below is what a JNI shim for the reference would bind
 * (see INTEGRATION.md for the Java side).  Citations name the reference interface each function replaces;
 * abbreviations (paths under the reference tree):
 *   F: = main/boofcv-feature/src/main/java/boofcv/    I: = main/boofcv-ip/src/main/java/boofcv/
 *   T: = main/boofcv-types/src/main/java/boofcv/
 *
 * Conventions
 *  - plain C, no C++ or torch types; the caller owns every host buffer, the library owns device memory;
 *  - every function returns 0 (BHIP_OK) or a negative bhip_status and never throws or aborts;
 *    bhip_last_error(ctx) gives the message.  A Java shim turns a non-zero status into RuntimeException,
 *    which is the reference's own "override did not handle it, run the Java code" signal
 *    (I:alg/filter/convolve/BOverrideConvolveImage.java:53-62);
 *  - one bhip_ctx per host thread per device; calls on one ctx are serialised on one HIP stream
 *    (reference objects are not thread safe either: one instance per thread);
 *  - images are GrayF32: pixel (x,y) = data[startIndex + y*stride + x]  (T:struct/image/ImageBase.java:34-52),
 *    so sub-images (startIndex != 0, stride > width) work everywhere;
 *  - pointers named dev_* are device (HBM) addresses on the ctx's device; every other pointer is host memory;
 *  - there is no CPU fallback inside the library: without a usable GPU bhip_ctx_create fails;
 *  - handles may be destroyed in any order and more than once: bhip_ctx_destroy releases the device side of every bhip_surf created on
 *    that context (they become inert: every call on them returns BHIP_ERR_INVALID, bhip_surf_destroy then only frees the shell), a
 *    pointer that is not a live handle is refused with BHIP_ERR_INVALID, and once the process is exiting the destroy calls do nothing
 *    (a finaliser that runs after the HIP runtime has shut down is harmless).
 */
#ifndef BOOFHIP_H
#define BOOFHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
	BHIP_OK = 0,
	BHIP_ERR_INVALID = -1,     /* bad argument / shape (IllegalArgumentException in the reference) */
	BHIP_ERR_UNSUPPORTED = -2, /* configuration not implemented on the GPU: caller should use the Java path */
	BHIP_ERR_HIP = -3,         /* HIP runtime error (message has the hipError string) */
	BHIP_ERR_NOMEM = -4,
	BHIP_ERR_CAPACITY = -5     /* internal fixed-capacity list overflowed even after regrow */
} bhip_status;

typedef struct bhip_ctx bhip_ctx;
typedef struct bhip_surf bhip_surf;

/* ---- configuration structs: same field names and defaults as the reference's Config* classes ---- */

/* F:abst/feature/detect/interest/ConfigFastHessian.java:33-70 */
typedef struct {
	float detectThreshold;     /* 1 */
	int extractRadius;         /* 2 */
	int maxFeaturesPerScale;   /* -1; > 0: SelectNBestFeatures per level (order = the restated ddogleg QuickSelect, unpinned vs the real jar) */
	int initialSampleSize;     /* 1 */
	int initialSize;           /* 9 */
	int numberScalesPerOctave; /* 4 */
	int numberOfOctaves;       /* 4 */
	int scaleStepSize;         /* 6 */
} bhip_fh_cfg;

/* F:abst/feature/describe/ConfigSurfDescribe.java:34-78 (Speed and Stability merged; useHaar=false only) */
typedef struct {
	int widthLargeGrid;    /* 4 */
	int widthSubRegion;    /* 5 */
	int widthSample;       /* 3 */
	double weightSigma;    /* Speed: 4.5 */
	int overLap;           /* Stability: 2 */
	double sigmaLargeGrid; /* Stability: 2.5 */
	double sigmaSubRegion; /* Stability: 2.5 */
} bhip_surf_cfg;

/* F:abst/feature/orientation/ConfigSlidingIntegral.java:34-54 (stable) /
 * ConfigAverageIntegral.java:34-51 (fast; windowSize ignored) */
typedef struct {
	double objectRadiusToScale; /* 1/BoofDefaults.SURF_SCALE_TO_RADIUS = 0.5 */
	double samplePeriod;        /* sliding 0.65, average 1 */
	double windowSize;          /* sliding pi/3 */
	int radius;                 /* sliding 8, average 6 */
	double weightSigma;         /* -1 */
	int sampleWidth;            /* 6 */
} bhip_ori_cfg;

void bhip_fh_cfg_default(bhip_fh_cfg* c);
void bhip_surf_cfg_default(bhip_surf_cfg* c);
void bhip_ori_cfg_default(bhip_ori_cfg* c, int stable);

/* ---- context ---- */
int bhip_ctx_create(int device, bhip_ctx** out);
/* same, but run on an existing hipStream_t (e.g. torch's current stream) instead of a private one */
int bhip_ctx_create_on_stream(int device, void* hip_stream, bhip_ctx** out);
int bhip_ctx_destroy(bhip_ctx* ctx);
int bhip_ctx_synchronize(bhip_ctx* ctx);
const char* bhip_last_error(bhip_ctx* ctx);
/* Page-locked ("pinned") host memory for the arrays a caller exchanges with the library -- what a JNI provider wraps in direct
 * ByteBuffers (NewDirectByteBuffer) for frames, fetched results and descriptor lists: copies to and from it are DMA transfers, copies
 * from ordinary (pageable) arrays go through the runtime's staging buffer.  Every entry point accepts either kind.  The block is not
 * tied to the context's lifetime; release it with bhip_host_free (a no-op for NULL and once process exit has begun). */
int bhip_host_alloc(bhip_ctx* ctx, long long bytes, uint8_t** host_mem);
int bhip_host_free(void* host_mem);
const char* bhip_version(void);
/* optional per-kernel timing with HIP events on the ctx stream (bench.py's live roofline numbers).  report writes one text line per
 * kernel tag, "tag launches total_ms algorithmic_bytes algorithmic_flops", and returns the buffer size it needs. */
int bhip_profile_enable(bhip_ctx* ctx, int on);
int bhip_profile_reset(bhip_ctx* ctx);
int bhip_profile_report(bhip_ctx* ctx, char* out, int cap);

/* ---- detect + describe: FactoryDetectDescribe.surfStable / surfFast -> DetectDescribePoint<GrayF32,BrightFeature>
 *      (F:factory/feature/detdesc/FactoryDetectDescribe.java:118-135,209-226; F:abst/feature/detdesc/DetectDescribePoint.java:32-46;
 *       F:abst/feature/detdesc/WrapDetectDescribeSurf.java:93-159).  NULL config = reference defaults. ---- */
int bhip_surf_create(bhip_ctx* ctx, const bhip_fh_cfg* fh, const bhip_surf_cfg* surf, const bhip_ori_cfg* ori, int stable, bhip_surf** out);
int bhip_surf_destroy(bhip_surf* s);
/* detect(T input) on a batch of host images (batch = 1 is the reference call).  Results are recycled by the next detect,
 * as in the reference (DetectDescribePoint.java:38-40). */
int bhip_surf_detect_f32(bhip_surf* s, const float* const* img, const int* startIndex, const int* stride, int width, int height, int batch);
/* same on a device-resident batch: image i starts at dev_images + i*imageStride floats, rows are `stride` floats apart.
 * Asynchronous on the ctx stream apart from one small count read-back: the call returns when the key points are counted (the frames have
 * been consumed by then), with the describe kernels still queued; counts are valid at once, every fetch waits for the results, device
 * views (bhip_surf_dev_view) and the resident associations are ordered behind them on the same stream. */
int bhip_surf_detect_dev_f32(bhip_surf* s, const float* dev_images, long long imageStride, int stride, int width, int height, int batch);
/* The same on GrayU8 frames: the integral images are GrayS32 (GIntegralImageOps.getIntegralType) and every stage runs on integer taps --
 * IntegralImageOps.transform(GrayU8, GrayS32), FastHessianFeatureDetector<GrayS32>, SparseIntegralGradient_NoBorder_I32 for the
 * orientation and the descriptor, convolveSparse(GrayS32) for the Laplacian sign.  Results through the same count / fetch calls;
 * bhip_surf_fetch_integral then returns the int32 words. */
int bhip_surf_detect_u8(bhip_surf* s, const uint8_t* const* img, const int* startIndex, const int* stride, int width, int height, int batch);
/* FactoryDetectDescribe.surfColorStable / surfColorFast (F:factory/feature/detdesc/FactoryDetectDescribe.java:154-176,246-268) on one
 * Planar<GrayF32> frame given as numBands band pointers of one shape: SurfPlanar_to_DetectDescribePoint.detect
 * (F:abst/feature/detdesc/SurfPlanar_to_DetectDescribePoint.java:62-77) = band average -> integral images of the average and of every band
 * -> Fast-Hessian on the average -> DetectDescribeSurfPlanar.describe (F:alg/feature/detdesc/DetectDescribeSurfPlanar.java:110-124;
 * orientation object radius = scale) -> DescribePointSurfPlanar.describe (F:alg/feature/describe/DescribePointSurfPlanar.java:100-114;
 * bands concatenated, normalised once; Laplacian sign from the average).  Results through bhip_surf_count / _fetch / _dev_view with
 * image = 0; bhip_surf_dof() then returns numBands * 64.  getRadius(i) of this wrapper is the scale itself (no factor 2). */
int bhip_surf_detect_planar_f32(bhip_surf* s, const float* const* bands, int numBands, int startIndex, int stride, int width, int height);
/* getNumberOfFeatures() of image `image` of the last detect */
int bhip_surf_count(bhip_surf* s, int image, int* n);
/* the same for every image of the last batch in one call: counts[i] for i < batch (capacity >= batch) */
int bhip_surf_counts(bhip_surf* s, int* counts, int capacity);
/* getLocation(i)/scale -> xy_scale[3n] ; getOrientation(i) -> angle[n] ; BrightFeature.white -> white[n] ;
 * getDescription(i).value -> desc[64n].  getRadius(i) = scale*2 (BoofDefaults.SURF_SCALE_TO_RADIUS).  Any pointer may be NULL. */
int bhip_surf_fetch(bhip_surf* s, int image, double* xy_scale, double* angle, uint8_t* white, double* desc);
/* the same for the WHOLE batch of the last detect in one set of copies: image i's slice starts at the exclusive prefix of the counts
 * (key point k of image i at index sum(count[0..i)) + k); arrays sized with bhip_surf_total */
int bhip_surf_fetch_all(bhip_surf* s, double* xy_scale, double* angle, uint8_t* white, double* desc);
/* AssociateDescription.associate() on descriptor lists that are still resident from the last detect of `s` (F:abst/feature/associate/
 * AssociateDescription.java:42-61 with lists a provider recognises as its own getDescription() objects): problem p associates image
 * srcImage[p] (source) with image dstImage[p] (destination), ScoreAssociateEuclideanSq_F64, same rules and results as bhip_assoc_l2_f64, no
 * descriptor upload.  pairs / fit: host arrays of bhip_surf_total entries; problem p's results start at the exclusive prefix of the counts
 * of srcImage[p] (an image may be the source of one problem per call); entries of images that are no source read -1 / 0.0. */
int bhip_assoc_l2_surf(bhip_surf* s, int count, const int* srcImage, const int* dstImage, double maxErr, int backwards, int* pairs, double* fit);

/* ---- Fast-Hessian + BRIEF as one DetectDescribePoint<T,TupleDesc_B>:
 *      FactoryDetectDescribe.fuseTogether(FactoryInterestPoint.fastHessian(fh), null, FactoryDescribeRegionPoint.brief(config, imageType))
 *      (F:factory/feature/detdesc/FactoryDetectDescribe.java:279-284; F:abst/feature/detdesc/DetectDescribeFusion.java:95-127;
 *       F:abst/feature/detect/interest/WrapFHtoInterestPoint.java:47-79; F:factory/feature/describe/FactoryDescribeRegionPoint.java:187-202;
 *       F:abst/feature/describe/WrapDescribeBrief.java:47-58; F:alg/feature/describe/DescribePointBrief.java:71-89), config.fixed = true.
 *      The definition (radius, numPoints, samplePoints, compare: FactoryBriefDefinition.gaussian2(new Random(123), radius, numPoints)) is
 *      generated on the Java side, as for bhip_brief_f32.  The object is a bhip_surf: detect with bhip_surf_detect_f32 / _dev_f32 / _u8 (GrayU8
 *      frames: GrayS32 integral image + ImplDescribeBinaryCompare_U8), read locations with bhip_surf_count / _total / _fetch (desc = NULL;
 *      getOrientation(i) is 0, getRadius(i) = scale*2), destroy with bhip_surf_destroy.  Every detected point is described (process() of the
 *      fixed BRIEF always returns true), in detector order; the words are taken from the frame itself (the reference blurs a copy it never
 *      reads), with the border rule of the image type (see bhip_brief_f32 / bhip_brief_u8). ---- */
int bhip_surf_create_brief(bhip_ctx* ctx, const bhip_fh_cfg* fh, int radius, int numPoints, const int32_t* samplePoints, const int32_t* compare,
						   bhip_surf** out);
/* getDescription(i).data (TupleDesc_B: ceil(numPoints/32) ints per feature, T:struct/feature/TupleDesc_B.java) of every feature of image
 * `image` of the last detect, or of the whole batch when image = -1 (image i's slice then starts at the exclusive prefix of the counts) */
int bhip_surf_fetch_brief(bhip_surf* s, int image, int32_t* words);
/* device view of the same words (valid until the next detect on s); *words = ints per feature */
int bhip_surf_dev_view_brief(bhip_surf* s, int image, const int32_t** dev_words, int* words, int* n);
/* AssociateDescription<TupleDesc_B>.associate() with ScoreAssociateHamming_B (F:alg/descriptor/DescriptorDistance.java:196-220) on the words still
 * resident from the last detect of a BRIEF object: contract of bhip_assoc_l2_surf, rules and results of bhip_assoc_hamming, no upload */
int bhip_assoc_hamming_surf(bhip_surf* s, int count, const int* srcImage, const int* dstImage, double maxErr, int backwards, int* pairs, double* fit);
/* device views of the same results (valid until the next detect): descriptors [n][dof] doubles, laplacian signs [n] bytes */
int bhip_surf_dev_view(bhip_surf* s, int image, const double** dev_desc, const double** dev_xy_scale, const uint8_t** dev_white, int* n);
int bhip_surf_dof(bhip_surf* s);
/* total key points over the whole batch of the last detect */
int bhip_surf_total(bhip_surf* s, long long* n);

/* stage-level entry points of the detector (used by the parity tests and by the BOverride nonmax hook) */
/* describe externally supplied points (x,y,scale) on the integral image of image `image` of the last detect:
 * WrapDetectDescribeSurf.computeDescriptors (:116-128) for a caller-provided foundPoints list */
int bhip_surf_describe_points(bhip_surf* s, int image, const double* xy_scale, int n, double* angle, uint8_t* white, double* desc);
/* copy the integral image of image `image` of the last detect to host (width*height floats, dense) */
int bhip_surf_fetch_integral(bhip_surf* s, int image, float* out);

/* GIntegralImageOps.transform -> ImplIntegralImageOps.transform(GrayF32,GrayF32) (I:alg/transform/ii/impl/ImplIntegralImageOps.java:42-66) */
int bhip_integral_f32(bhip_ctx* ctx, const float* in, int inStart, int inStride, int width, int height, float* out, int outStart, int outStride);
/* IntegralImageFeatureIntensity.hessian(GrayF32,skip,size,GrayF32) (F:alg/feature/detect/intensity/IntegralImageFeatureIntensity.java:43-56);
 * intensity is (width/skip) x (height/skip) */
int bhip_hessian_f32(bhip_ctx* ctx, const float* ii, int iiStart, int iiStride, int width, int height, int skip, int size, float* intensity,
					 int outStart, int outStride);
/* NonMaxSuppression.process for the strict block extractor built by FactoryFeatureExtractor.nonmax(ConfigExtract(radius,threshold,border,true))
 * (F:factory/feature/detect/extract/FactoryFeatureExtractor.java:63-102; F:alg/feature/detect/extract/NonMaxBlock.java:69-94).
 * Writes up to cap (x,y) int16 pairs in block-raster order (the USE_CONCURRENT=false order); *n is the number found. */
int bhip_nonmax_block_f32(bhip_ctx* ctx, const float* intensity, int start, int stride, int width, int height, int radius, float threshold,
						  int border, int16_t* xy, int cap, int* n);
/* SelectNBestFeatures.process(intensity, corners, positive) + getBestCorners() (F:alg/feature/detect/extract/SelectNBestFeatures.java:51-97):
 * n <= target copies the list; otherwise keys = -intensity (positive) or +intensity and org.ddogleg.sorting.QuickSelect.selectIndex(keys,
 * target, n, indexes) decides which `target` corners are kept and in which order.  ddogleg is not part of the reference tree: the routine
 * is the published Numerical Recipes `select` with an index array (see oracle/boof_oracle.hpp quickSelectIndex); the kept SET is pinned
 * (the N most intense, exact ties at the cut aside), the order is "parity unpinned".  xy / out_xy: (x,y) int16 pairs; out_xy holds
 * min(n, target) pairs.  Used by FastHessianFeatureDetector (maxFeaturesPerScale > 0) and GeneralFeatureDetector (maxFeatures > 0). */
int bhip_select_nbest_f32(bhip_ctx* ctx, const float* intensity, int start, int stride, int width, int height, const int16_t* xy, int n, int target,
						  int positive, int16_t* out_xy, int* out_n);
/* FastHessianFeatureDetector.detect(ii) (F:alg/feature/detect/interest/FastHessianFeatureDetector.java:156-188) on a host integral image */
int bhip_fh_detect_f32(bhip_ctx* ctx, const bhip_fh_cfg* cfg, const float* ii, int iiStart, int iiStride, int width, int height, double* xy_scale,
					   int cap, int* n);

/* ---- association: FactoryAssociation.greedy(score,maxErr,backwards) -> AssociateDescription
 *      (F:factory/feature/associate/FactoryAssociation.java:51-65; F:alg/feature/associate/AssociateGreedy.java:65-118;
 *       F:abst/feature/associate/WrapAssociateGreedy.java:73-93).  pairs[i] = dst index or -1, fit[i] = AssociateGreedyBase.fitQuality. ---- */
/* ScoreAssociateEuclideanSq_F64 (DescriptorDistance.euclideanSq, F:alg/descriptor/DescriptorDistance.java:55-64); sqrtScore!=0 gives
 * ScoreAssociateEuclidean_F64 (:36-46) */
int bhip_assoc_l2_f64(bhip_ctx* ctx, const double* src, int ns, const double* dst, int nd, int dof, double maxErr, int backwards, int sqrtScore,
					  int* pairs, double* fit);
/* ScoreAssociateHamming_B (DescriptorDistance.hamming, :196-220) on TupleDesc_B.data words */
int bhip_assoc_hamming(bhip_ctx* ctx, const int32_t* src, int ns, const int32_t* dst, int nd, int words, double maxErr, int backwards, int* pairs,
					   double* fit);
/* device-resident forms (async on the ctx stream; outputs are device arrays) */
int bhip_assoc_l2_dev(bhip_ctx* ctx, const double* dev_src, int ns, const double* dev_dst, int nd, int dof, double maxErr, int backwards,
					  int sqrtScore, int* dev_pairs, double* dev_fit);
int bhip_assoc_hamming_dev(bhip_ctx* ctx, const int32_t* dev_src, int ns, const int32_t* dev_dst, int nd, int words, double maxErr, int backwards,
						   int* dev_pairs, double* dev_fit);
/* batched device form: `count` independent (src,dst) problems in one launch sequence.  Problem p uses rows
 * [srcOff[p], srcOff[p]+ns[p]) of dev_src and [dstOff[p], dstOff[p]+nd[p]) of dev_dst (host arrays of offsets/sizes);
 * pairs/fit are written at the source row offsets. */
int bhip_assoc_l2_dev_batched(bhip_ctx* ctx, const double* dev_src, const double* dev_dst, int dof, int count, const long long* srcOff,
							  const int* ns, const long long* dstOff, const int* nd, double maxErr, int backwards, int* dev_pairs, double* dev_fit);
/* the same for ScoreAssociateHamming_B word lists (`words` ints per row): the consecutive-frame problems of a batch of BRIEF frames */
int bhip_assoc_hamming_dev_batched(bhip_ctx* ctx, const int32_t* dev_src, const int32_t* dev_dst, int words, int count, const long long* srcOff,
								   const int* ns, const long long* dstOff, const int* nd, double maxErr, int backwards, int* dev_pairs, double* dev_fit);

/* sharded association: this rank owns source rows [srcBegin, srcBegin+nsLocal) of a global problem with nsGlobal rows and the
 * whole destination set.  Phase 1 computes the local forward matches and, per destination column, the local column top-2
 * (min1, argmin1 as GLOBAL source index, min2) into dev_colTop (nd records of {double min1; double min2; int idx1; int pad}).
 * The caller all-gathers dev_colTop across ranks (RCCL, e.g. torch.distributed.all_gather_into_tensor) into nranks*nd records;
 * phase 2 merges them and applies the strict column-minimum rule to the local rows. */
int bhip_assoc_l2_shard_phase1(bhip_ctx* ctx, const double* dev_src, int nsLocal, int srcBegin, const double* dev_dst, int nd, int dof,
							   double maxErr, int* dev_pairs, double* dev_fit, void* dev_colTop);
int bhip_assoc_hamming_shard_phase1(bhip_ctx* ctx, const int32_t* dev_src, int nsLocal, int srcBegin, const int32_t* dev_dst, int nd, int words,
									double maxErr, int* dev_pairs, double* dev_fit, void* dev_colTop);
int bhip_assoc_shard_phase2(bhip_ctx* ctx, const void* dev_colTopAll, int nranks, int nd, int nsLocal, int srcBegin, int* dev_pairs,
							double* dev_fit);
int bhip_assoc_coltop_bytes(void); /* sizeof one column record */

/* ---- boofcv-ip front end behind the BOverride* hooks ---- */
/* BOverrideConvolveImage.horizontal/vertical (I:alg/filter/convolve/BOverrideConvolveImage.java:37-51) = ConvolveImageNoBorder
 * (I:alg/filter/convolve/ConvolveImageNoBorder.java:53-77): border pixels of out are left untouched */
int bhip_conv_h_f32(bhip_ctx* ctx, const float* kernel, int kernelWidth, int kernelOffset, const float* in, int inStart, int inStride, int width,
					int height, float* out, int outStart, int outStride);
int bhip_conv_v_f32(bhip_ctx* ctx, const float* kernel, int kernelWidth, int kernelOffset, const float* in, int inStart, int inStride, int width,
					int height, float* out, int outStart, int outStride);
/* BOverrideConvolveImageNormalized.horizontal/vertical (I:alg/filter/convolve/BOverrideConvolveImageNormalized.java:38-52) =
 * ConvolveImageNormalized.horizontal/vertical (I:alg/filter/convolve/ConvolveImageNormalized.java:48-93) */
int bhip_conv_norm_h_f32(bhip_ctx* ctx, const float* kernel, int kernelWidth, int kernelOffset, const float* in, int inStart, int inStride,
						 int width, int height, float* out, int outStart, int outStride);
int bhip_conv_norm_v_f32(bhip_ctx* ctx, const float* kernel, int kernelWidth, int kernelOffset, const float* in, int inStart, int inStride,
						 int width, int height, float* out, int outStart, int outStride);
/* BOverrideBlurImageOps.gaussian (I:alg/filter/blur/BOverrideBlurImageOps.java:38,48-49) = BlurImageOps.gaussian(GrayF32,out,sigma,radius,storage)
 * (I:alg/filter/blur/BlurImageOps.java:406-425), sigmaX==sigmaY / radiusX==radiusY form */
int bhip_gaussian_f32(bhip_ctx* ctx, const float* in, int inStart, int inStride, int width, int height, double sigma, int radius, float* out,
					  int outStart, int outStride);
/* BOverrideConvolveImage.convolve target: ConvolveImageNoBorder.convolve(Kernel2D_F32, GrayF32, GrayF32)
 * (I:alg/filter/convolve/ConvolveImageNoBorder.java:79-90; unrolled widths 3..11 sum every kernel row from 0 and add the row sums,
 * I:alg/filter/convolve/noborder/ConvolveImageUnrolled_SB_F32_F32.java:592-644; otherwise ConvolveImageStandard_SB.java:106-134).
 * kernel = kernelWidth x kernelWidth values, row-major; the frame of `out` is left untouched. */
int bhip_conv2d_f32(bhip_ctx* ctx, const float* kernel, int kernelWidth, int kernelOffset, const float* in, int inStart, int inStride, int width,
					int height, float* out, int outStart, int outStride);
/* BOverrideBlurImageOps.mean target: BlurImageOps.mean(GrayF32, out, radiusX, radiusY, storage) (I:alg/filter/blur/BlurImageOps.java:359-376)
 * = ConvolveImageMean.horizontal then vertical (I:alg/filter/convolve/ConvolveImageMean.java:55-101): float running sums in the
 * single-threaded order of ImplConvolveMean (I:alg/filter/convolve/noborder/ImplConvolveMean.java:281-357), re-normalised border. */
int bhip_mean_f32(bhip_ctx* ctx, const float* in, int inStart, int inStride, int width, int height, int radiusX, int radiusY, float* out,
				  int outStart, int outStride);
/* BOverrideBlurImageOps.median target: BlurImageOps.median(GrayF32, out, radius) (I:alg/filter/blur/BlurImageOps.java:752-765) =
 * ImplMedianSortNaive.process (I:alg/filter/blur/impl/ImplMedianSortNaive.java:97-135): the (count/2)-th order statistic of the clipped window */
int bhip_median_f32(bhip_ctx* ctx, const float* in, int inStart, int inStride, int width, int height, int radius, float* out, int outStart,
					int outStride);
/* GradientSobel.process(GrayF32,derivX,derivY,border) (I:alg/filter/derivative/GradientSobel.java:158-173);
 * border: 0 = null (frame untouched), 1 = ImageBorderValue(0) */
int bhip_sobel_f32(bhip_ctx* ctx, const float* in, int inStart, int inStride, int width, int height, float* dx, float* dy, int outStart,
				   int outStride, int border);
/* GradientThree.process(GrayF32,...) -> GradientThree_Standard.process (I:alg/filter/derivative/impl/GradientThree_Standard.java:40-62) */
int bhip_three_f32(bhip_ctx* ctx, const float* in, int inStart, int inStride, int width, int height, float* dx, float* dy, int outStart,
				   int outStride, int border);
/* ConvolveImageDownNormalized.horizontal/vertical (I:alg/filter/convolve/ConvolveImageDownNormalized.java:53-86): every skip-th pixel
 * along the filtered axis, kernel re-normalised where it overlaps the border (ConvolveDownNormalized_JustBorder.java:43-139), plain
 * sum inside (ConvolveDownNoBorderUnrolled_F32_F32 / ConvolveDownNoBorderStandard), naive form when kernelWidth >= width.  `out` is
 * outWidth x outHeight and must satisfy ConvolveImageDownNoBorder.checkParametersH/V (:160-176); pixels the reference does not write
 * keep the caller's values.  BHIP_ERR_INVALID where the reference throws. */
int bhip_conv_down_norm_h_f32(bhip_ctx* ctx, const float* kernel, int kernelWidth, const float* in, int inStart, int inStride, int width, int height,
							  float* out, int outStart, int outStride, int outWidth, int outHeight, int skip);
int bhip_conv_down_norm_v_f32(bhip_ctx* ctx, const float* kernel, int kernelWidth, const float* in, int inStart, int inStride, int width, int height,
							  float* out, int outStart, int outStride, int outWidth, int outHeight, int skip);
/* PyramidDiscreteSampleBlur (I:alg/transform/pyramid/PyramidDiscreteSampleBlur.java:68-126).  bhip_pyramid_layout is
 * ImagePyramidBase.initialize + checkScales (T:struct/pyramid/ImagePyramidBase.java:73-112): dims[2i],dims[2i+1] = width,height of layer i,
 * offsets[i] = first float of layer i in the packed output (layers dense, stride = layer width), *totalFloats = floats per frame.
 * bhip_pyramid_f32 = process(input) for one host image; bhip_pyramid_dev_f32 runs `batch` device-resident frames (frame b of the
 * output starts at dev_out + b * totalFloats) without leaving the stream.  The 1-D kernel is FactoryPyramid.discreteGaussian's
 * FactoryKernelGaussian.gaussian(Kernel1D_F32, sigma, radius) (I:factory/transform/pyramid/FactoryPyramid.java:53-61), built by the caller. */
/* FactoryKernelGaussian.gaussian(Kernel1D_F32.class, sigma, radius) (I:factory/filter/kernel/FactoryKernelGaussian.java:120-153): host-only
 * helper for callers outside the JVM.  Returns the kernel width, or -(needed width) when capacity is too small / out is NULL. */
int bhip_gaussian_kernel1d_f32(double sigma, int radius, float* out, int capacity);
int bhip_pyramid_layout(int width, int height, const int* scales, int numLayers, int* dims, long long* offsets, long long* totalFloats);
int bhip_pyramid_f32(bhip_ctx* ctx, const float* kernel, int kernelWidth, const int* scales, int numLayers, const float* in, int inStart,
					 int inStride, int width, int height, float* out);
int bhip_pyramid_dev_f32(bhip_ctx* ctx, const float* kernel, int kernelWidth, const int* scales, int numLayers, const float* dev_in,
						 long long inImageStride, int inStride, int width, int height, int batch, float* dev_out);
/* FactoryIntensityPointAlg.shiTomasi(radius, false, GrayF32) / harris(radius, kappa, false, GrayF32)
 * (F:factory/feature/detect/intensity/FactoryIntensityPointAlg.java:91-160) -> GradientCornerIntensity.process(derivX, derivY, intensity)
 * = ImplSsdCorner_F32 (F:alg/feature/detect/intensity/impl/ImplSsdCorner_F32.java:62-196, ImplSsdCornerBox.java:36-51) with
 * ShiTomasiCorner_F32 (kind 0) or HarrisCorner_F32 (kind 1): box-window running sums of dx*dx, dx*dy, dy*dy in the reference's
 * single-threaded order (bit-exact), intensity 0 inside the border of `radius` pixels.  derivX / derivY share startIndex and stride.
 * Together with bhip_sobel_f32 and bhip_nonmax_block_f32 this is GeneralFeatureDetector.process for maxFeatures <= 0
 * (F:alg/feature/detect/interest/GeneralFeatureDetector.java:118-160). */
int bhip_corner_intensity_f32(bhip_ctx* ctx, int kind, int radius, float kappa, const float* derivX, const float* derivY, int dStart, int dStride,
							  int width, int height, float* intensity, int iStart, int iStride);
/* Integer image variants at stage level.
 * bhip_integral_u8_s32: IntegralImageOps.transform(GrayU8, GrayS32) (I:alg/transform/ii/impl/ImplIntegralImageOps.java:94-118).
 * bhip_hessian_s32: IntegralImageFeatureIntensity.hessian(GrayS32, skip, size, GrayF32) (F:alg/feature/detect/intensity/impl/
 *   ImplIntegralImageFeatureIntensity.java:245-390): integer box sums, converted to float where the Java code assigns them to a float.
 * bhip_brief_u8: DescribePointBrief on GrayU8 = ImplDescribeBinaryCompare_U8 (F:alg/feature/describe/impl/ImplDescribeBinaryCompare_U8.java:47-101;
 *   its border form shifts the word for every pair, the F32 class only for pairs inside the image). */
int bhip_integral_u8_s32(bhip_ctx* ctx, const uint8_t* in, int inStart, int inStride, int width, int height, int32_t* out, int outStart, int outStride);
int bhip_hessian_s32(bhip_ctx* ctx, const int32_t* ii, int iiStart, int iiStride, int width, int height, int skip, int size, float* out, int outStart,
					 int outStride);
/* FastHessianFeatureDetector<GrayS32>.detect(integral) (F:alg/feature/detect/interest/FastHessianFeatureDetector.java:156-188 on the integral image
 * of a GrayU8 frame): same outputs as bhip_fh_detect_f32 */
int bhip_fh_detect_s32(bhip_ctx* ctx, const bhip_fh_cfg* cfg, const int32_t* ii, int iiStart, int iiStride, int width, int height,
					   double* xy_scale, int cap, int* n);
int bhip_brief_u8(bhip_ctx* ctx, const uint8_t* img, int start, int stride, int width, int height, int radius, int numPoints,
				  const int32_t* samplePoints, const int32_t* compare, const double* xy, int n, int32_t* out);
/* DescribePointBrief.process for n points on one image (F:alg/feature/describe/DescribePointBrief.java:73-89;
 * F:alg/feature/describe/impl/ImplDescribeBinaryCompare_F32.java:47-101).  The definition (samplePoints[numPoints][2], compare[numPoints][2])
 * is supplied by the caller: FactoryBriefDefinition.gaussian2 depends on java.util.Random + StrictMath and is generated on the Java side. */
int bhip_brief_f32(bhip_ctx* ctx, const float* img, int start, int stride, int width, int height, int radius, int numPoints,
				   const int32_t* samplePoints, const int32_t* compare, const double* xy, int n, int32_t* out);

/* ---- device-resident, batched forms of the boofcv-ip front end (BASELINE config 5: pyramid -> gradient -> non-max -> SURF on a 4K stream that
 *      never leaves HBM).  Image b of a batch starts imageStride floats after image 0, rows are `stride` floats apart; calls are asynchronous on
 *      the ctx stream.  Same kernels and arithmetic as the host-buffer entry points above (each cites the same reference function). ---- */
/* ConvolveImageNoBorder.horizontal / vertical (I:alg/filter/convolve/ConvolveImageNoBorder.java:53-77): frame of dev_out untouched */
int bhip_conv_h_dev_f32(bhip_ctx* ctx, const float* kernel, int kernelWidth, int kernelOffset, const float* dev_in, long long inImageStride, int inStride,
						int width, int height, int batch, float* dev_out, long long outImageStride, int outStride);
int bhip_conv_v_dev_f32(bhip_ctx* ctx, const float* kernel, int kernelWidth, int kernelOffset, const float* dev_in, long long inImageStride, int inStride,
						int width, int height, int batch, float* dev_out, long long outImageStride, int outStride);
/* ConvolveImageNormalized.horizontal / vertical (I:alg/filter/convolve/ConvolveImageNormalized.java:48-93) */
int bhip_conv_norm_h_dev_f32(bhip_ctx* ctx, const float* kernel, int kernelWidth, int kernelOffset, const float* dev_in, long long inImageStride, int inStride,
							 int width, int height, int batch, float* dev_out, long long outImageStride, int outStride);
int bhip_conv_norm_v_dev_f32(bhip_ctx* ctx, const float* kernel, int kernelWidth, int kernelOffset, const float* dev_in, long long inImageStride, int inStride,
							 int width, int height, int batch, float* dev_out, long long outImageStride, int outStride);
/* BlurImageOps.gaussian(GrayF32, out, sigma, radius, storage) (I:alg/filter/blur/BlurImageOps.java:406-425); `storage` is library scratch */
int bhip_gaussian_dev_f32(bhip_ctx* ctx, const float* dev_in, long long inImageStride, int inStride, int width, int height, int batch, double sigma, int radius,
						  float* dev_out, long long outImageStride, int outStride);
/* GradientSobel.process (I:alg/filter/derivative/GradientSobel.java:158-173) / GradientThree.process -> GradientThree_Standard
 * (I:alg/filter/derivative/impl/GradientThree_Standard.java:40-62); dx and dy share outImageStride / outStride; border as in bhip_sobel_f32 */
int bhip_sobel_dev_f32(bhip_ctx* ctx, const float* dev_in, long long inImageStride, int inStride, int width, int height, int batch, float* dev_dx, float* dev_dy,
					   long long outImageStride, int outStride, int border);
int bhip_three_dev_f32(bhip_ctx* ctx, const float* dev_in, long long inImageStride, int inStride, int width, int height, int batch, float* dev_dx, float* dev_dy,
					   long long outImageStride, int outStride, int border);
/* Gradient magnitude images: kind 0 = GradientToEdgeFeatures.intensityE, (float)Math.sqrt(dx*dx + dy*dy); kind 1 = intensityAbs, |dx| + |dy|
 * (F:alg/feature/detect/edge/GradientToEdgeFeatures.java:61-95 -> impl/ImplGradientToEdgeFeatures.java:40-85); kind 2 = dx*dx + dy*dy, the
 * |grad|^2 image BASELINE config 5 runs the non-max suppression on (the products and the sum of intensityE without the root) */
int bhip_gradient_intensity_dev_f32(bhip_ctx* ctx, int kind, const float* dev_dx, const float* dev_dy, long long dImageStride, int dStride, int width,
									int height, int batch, float* dev_out, long long outImageStride, int outStride);
/* NonMaxBlock.process, strict rule (F:alg/feature/detect/extract/NonMaxBlock.java:69-94; NonMaxBlockSearchStrict.java:56-79,196-221) on every image of
 * a batch: image b's maxima go to dev_xy[b*cap ...] as (x,y) int16 pairs in block-raster order, their number to dev_n[b] (it may exceed cap;
 * only the first cap pairs are written) */
int bhip_nonmax_block_dev_f32(bhip_ctx* ctx, const float* dev_intensity, long long imageStride, int stride, int width, int height, int batch, int radius,
							  float threshold, int border, int16_t* dev_xy, int cap, int* dev_n);
/* GradientCornerIntensity.process (see bhip_corner_intensity_f32) on a batch; derivX / derivY share dImageStride / dStride */
int bhip_corner_intensity_dev_f32(bhip_ctx* ctx, int kind, int radius, float kappa, const float* dev_dx, const float* dev_dy, long long dImageStride,
								  int dStride, int width, int height, int batch, float* dev_intensity, long long iImageStride, int iStride);
/* DescribePointBrief.process (see bhip_brief_f32) for the points of a batch: image b owns points [start[b], start[b+1]) of dev_xy ((x,y) doubles;
 * `start` is a host array of batch+1 entries); point p's words go to dev_out[p * ceil(numPoints/32) ...] */
int bhip_brief_dev_f32(bhip_ctx* ctx, const float* dev_img, long long imageStride, int stride, int width, int height, int batch, int radius, int numPoints,
					   const int32_t* samplePoints, const int32_t* compare, const double* dev_xy, const int* start, int32_t* dev_out);

#ifdef __cplusplus
}
#endif
#endif /* BOOFHIP_H */
