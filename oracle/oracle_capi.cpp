// TEST INFRASTRUCTURE ONLY -- C entry points of the CPU oracle (see boof_oracle.hpp header comment).
// Loaded through ctypes by oracle/pyoracle.py from tests/, __graft_entry__.smoke() and bench.py's
// cpu_baseline leg.  The product (boofcv_amd/) never links or loads this library.
#include "boof_oracle.hpp"
#include "boof_oracle_ip.hpp"
#include "boof_oracle_int.hpp"
#include <omp.h>

using namespace oracle;

extern "C" {

struct orc_image {
	float* data;
	int startIndex, stride, width, height;
};
static inline GrayF32 view(const orc_image* im) { return GrayF32(im->data, im->startIndex, im->stride, im->width, im->height); }

struct orc_fh_cfg {
	float detectThreshold;
	int extractRadius, maxFeaturesPerScale, initialSampleSize, initialSize, numberScalesPerOctave, numberOfOctaves, scaleStepSize;
};
struct orc_surf_cfg {
	int widthLargeGrid, widthSubRegion, widthSample;
	double weightSigma;
	int overLap;
	double sigmaLargeGrid, sigmaSubRegion;
};
struct orc_ori_cfg {
	double objectRadiusToScale, samplePeriod, windowSize;
	int radius;
	double weightSigma;
	int sampleWidth;
};

static ConfigFastHessian toFh(const orc_fh_cfg* c) {
	ConfigFastHessian r;
	if (!c) return r;
	r.detectThreshold = c->detectThreshold; r.extractRadius = c->extractRadius; r.maxFeaturesPerScale = c->maxFeaturesPerScale;
	r.initialSampleSize = c->initialSampleSize; r.initialSize = c->initialSize; r.numberScalesPerOctave = c->numberScalesPerOctave;
	r.numberOfOctaves = c->numberOfOctaves; r.scaleStepSize = c->scaleStepSize;
	return r;
}
static ConfigSurfDescribe toSd(const orc_surf_cfg* c) {
	ConfigSurfDescribe r;
	if (!c) return r;
	r.widthLargeGrid = c->widthLargeGrid; r.widthSubRegion = c->widthSubRegion; r.widthSample = c->widthSample;
	r.weightSigma = c->weightSigma; r.overLap = c->overLap; r.sigmaLargeGrid = c->sigmaLargeGrid; r.sigmaSubRegion = c->sigmaSubRegion;
	return r;
}
static ConfigSlidingIntegral toSliding(const orc_ori_cfg* c) {
	ConfigSlidingIntegral r;
	if (!c) return r;
	r.objectRadiusToScale = c->objectRadiusToScale; r.samplePeriod = c->samplePeriod; r.windowSize = c->windowSize;
	r.radius = c->radius; r.weightSigma = c->weightSigma; r.sampleWidth = c->sampleWidth;
	return r;
}
static ConfigAverageIntegral toAverage(const orc_ori_cfg* c) {
	ConfigAverageIntegral r;
	if (!c) return r;
	r.objectRadiusToScale = c->objectRadiusToScale; r.samplePeriod = c->samplePeriod;
	r.radius = c->radius; r.weightSigma = c->weightSigma; r.sampleWidth = c->sampleWidth;
	return r;
}

int orc_max_threads() { return omp_get_max_threads(); }

// ---- java.util.Random ----
void* orc_rand_new(int64_t seed) { return new JavaRandom(seed); }
void orc_rand_free(void* r) { delete (JavaRandom*)r; }
int32_t orc_rand_next_int(void* r) { return ((JavaRandom*)r)->nextInt(); }
int32_t orc_rand_next_int_bound(void* r, int32_t b) { return ((JavaRandom*)r)->nextInt(b); }
float orc_rand_next_float(void* r) { return ((JavaRandom*)r)->nextFloat(); }
double orc_rand_next_double(void* r) { return ((JavaRandom*)r)->nextDouble(); }
double orc_rand_next_gaussian(void* r) { return ((JavaRandom*)r)->nextGaussian(); }
int orc_rand_next_boolean(void* r) { return ((JavaRandom*)r)->nextBoolean(); }
void orc_fill_uniform(void* r, const orc_image* im, float min, float max) { GrayF32 v = view(im); fillUniform(v, *(JavaRandom*)r, min, max); }
void orc_fill_gaussian(void* r, const orc_image* im, double mean, double sigma, float lower, float upper) {
	GrayF32 v = view(im);
	fillGaussian(v, *(JavaRandom*)r, mean, sigma, lower, upper);
}

// ---- integral image ----
void orc_integral(const orc_image* in, const orc_image* out) { GrayF32 o = view(out); integral_transform(view(in), o); }
float orc_block_unsafe(const orc_image* ii, int x0, int y0, int x1, int y1) { return block_unsafe(view(ii), x0, y0, x1, y1); }
float orc_block_zero(const orc_image* ii, int x0, int y0, int x1, int y1) { return block_zero(view(ii), x0, y0, x1, y1); }
// kind: 0 = XX, 1 = YY, 2 = XY
float orc_convolve_sparse(const orc_image* ii, int kind, int size, int x, int y) {
	IntegralKernel k = kind == 0 ? kernelDerivXX(size) : kind == 1 ? kernelDerivYY(size) : kernelDerivXY(size);
	return convolveSparse(view(ii), k, x, y);
}
// generic kernel given as blocks (for the convolveSparse-vs-dense known-answer test)
float orc_convolve_sparse_blocks(const orc_image* ii, int n, const int* blocks /*4n*/, const int* scales, int x, int y) {
	IntegralKernel k; k.n = n;
	for (int i = 0; i < n; i++) k.set(i, blocks[4 * i], blocks[4 * i + 1], blocks[4 * i + 2], blocks[4 * i + 3], scales[i]);
	return convolveSparse(view(ii), k, x, y);
}

// ---- hessian intensity ---- variant: 0 = border+inner (the production path), 1 = naive
void orc_hessian(const orc_image* ii, int skip, int size, const orc_image* intensity, int variant, int threads) {
	GrayF32 o = view(intensity);
	if (variant == 1) hessianNaive(view(ii), skip, size, o);
	else hessian(view(ii), skip, size, o, threads);
}

// ---- non-max ---- variant: 0 = block algorithm (block-raster order), 1 = naive
int orc_nonmax(const orc_image* intensity, int radius, float threshold, int border, int variant, int16_t* out_xy, int cap, int threads) {
	QueueCorner q;
	GrayF32 img = view(intensity);
	if (variant == 1) nonmaxNaiveStrict(img, radius, threshold, border, q);
	else {
		NonMaxBlockStrictMax nm; nm.radius = radius; nm.thresholdMax = threshold; nm.border = border;
		nm.process(img, q, threads);
	}
	int n = (int)q.size();
	for (int i = 0; i < n && i < cap; i++) { out_xy[2 * i] = q[i].x; out_xy[2 * i + 1] = q[i].y; }
	return n;
}

// ---- SelectNBestFeatures.process(intensity, corners, positive): out_xy gets min(n, target) points in the reference's output order ----
int orc_select_nbest(const orc_image* intensity, const int16_t* xy, int n, int target, int positive, int16_t* out_xy) {
	QueueCorner q, best;
	for (int i = 0; i < n; i++) q.push_back(Point2D_I16{xy[2 * i], xy[2 * i + 1]});
	FastHessianFeatureDetector::selectNBest(view(intensity), q, target, best, positive != 0);
	for (size_t i = 0; i < best.size(); i++) { out_xy[2 * i] = best[i].x; out_xy[2 * i + 1] = best[i].y; }
	return (int)best.size();
}

// ---- fast hessian detector on an integral image ----
int orc_fh_detect(const orc_image* ii, const orc_fh_cfg* cfg, double* out_xys, int cap, int threads) {
	FastHessianFeatureDetector det(toFh(cfg));
	det.threads = threads;
	det.detect(view(ii));
	int n = (int)det.foundPoints.size();
	for (int i = 0; i < n && i < cap; i++) {
		out_xys[3 * i] = det.foundPoints[i].x; out_xys[3 * i + 1] = det.foundPoints[i].y; out_xys[3 * i + 2] = det.foundPoints[i].scale;
	}
	return n;
}

// ---- orientation ---- kind: 0 = sliding window, 1 = average
double orc_orientation(const orc_image* ii, int kind, const orc_ori_cfg* cfg, double x, double y, double objectRadius) {
	GrayF32 v = view(ii);
	if (kind == 0) {
		OrientationSlidingWindow o(toSliding(cfg));
		o.setImage(v); o.setObjectRadius(objectRadius);
		return o.compute(x, y);
	}
	OrientationAverage o(toAverage(cfg));
	o.setImage(v); o.setObjectRadius(objectRadius);
	return o.compute(x, y);
}

// ---- sparse gradient ----
int orc_sparse_gradient(const orc_image* ii, double width, int x, int y, float* gx, float* gy) {
	GrayF32 v = view(ii);
	SparseIntegralGradient_NoBorder_F32 g; g.input = &v; g.setWidth(width);
	if (!g.isInBounds(x, y)) { *gx = 0; *gy = 0; return 0; }
	g.compute(x, y, *gx, *gy);
	return 1;
}

// ---- descriptor ---- stable: 1 = DescribePointSurfMod, 0 = DescribePointSurf
void orc_describe(const orc_image* ii, int stable, const orc_surf_cfg* cfg, double x, double y, double angle, double scale, double* desc, uint8_t* white,
				  int normalize) {
	GrayF32 v = view(ii);
	BrightFeature bf;
	ConfigSurfDescribe c = toSd(cfg);
	if (stable) {
		DescribePointSurfMod d(c); d.setImage(v);
		if (normalize) d.describe(x, y, angle, scale, bf);
		else { bf.value.resize(d.featureDOF); d.describeTuple(x, y, angle, scale, bf.value.data()); }
	} else {
		DescribePointSurf d(c); d.setImage(v);
		if (normalize) d.describe(x, y, angle, scale, bf);
		else { bf.value.resize(d.featureDOF); d.describeTuple(x, y, angle, scale, bf.value.data()); }
	}
	std::memcpy(desc, bf.value.data(), sizeof(double) * bf.value.size());
	*white = bf.white ? 1 : 0;
}


// ---- small pieces re-expressed for the reference's own unit tests (tests/test_oracle_reference_tests.py) ----
// SurfDescribeOps.isInside(ii, X, Y, radiusRegions, kernelSize, scale, c, s)   F:alg/feature/describe/SurfDescribeOps.java:120-159
int orc_surf_is_inside(int width, int height, double X, double Y, int radiusRegions, int kernelSize, double scale, double c, double s) {
	GrayF32 v(width, height);
	return surfIsInside(v, X, Y, radiusRegions, kernelSize, scale, c, s) ? 1 : 0;
}
// SurfDescribeOps.isInside(width, height, tl_x, tl_y, regionSize, sampleSize)   :183-204
int orc_surf_is_inside_region(int width, int height, double tl_x, double tl_y, double regionSize, double sampleSize) {
	return surfIsInsideRegion(width, height, tl_x, tl_y, regionSize, sampleSize) ? 1 : 0;
}
// UtilFeature.normalizeL2(TupleDesc_F64)   F:alg/descriptor/UtilFeature.java:101-114
void orc_normalize_l2(double* v, int n) { normalizeL2(v, n); }
// org.ddogleg.stats.UtilGaussian.computePDF as restated (published formula; ddogleg is not in the reference tree)
double orc_compute_pdf(double mean, double sigma, double sample) { return computePDF(mean, sigma, sample); }
// SparseIntegralGradient_NoBorder_F32.setWidth / isInBounds: the sample box bounds and the in-bounds predicate
int orc_sparse_gradient_bounds(int width, int height, double kernelWidth, int x, int y, int* box /*x0,y0,x1,y1*/) {
	GrayF32 v(width, height);
	SparseIntegralGradient_NoBorder_F32 g; g.input = &v; g.setWidth(kernelWidth);
	box[0] = g.x0; box[1] = g.y0; box[2] = g.x1; box[3] = g.y1;
	return g.isInBounds(x, y) ? 1 : 0;
}

// ---- kernels (for table tests) ----
int orc_gaussian_width(double sigma, int width, double* out) {
	Kernel2D_F64 k = gaussianWidth(sigma, width);
	std::memcpy(out, k.data.data(), sizeof(double) * k.data.size());
	return k.width;
}
int orc_gaussian2d_f64(double sigma, int radius, double* out) {
	Kernel2D_F64 k = gaussian2D_F64_auto(sigma, radius);
	std::memcpy(out, k.data.data(), sizeof(double) * k.data.size());
	return k.width;
}
int orc_gaussian1d_f32(double sigma, int radius, float* out) {
	Kernel1D_F32 k = gaussian1D_F32(sigma, radius);
	std::memcpy(out, k.data.data(), sizeof(float) * k.data.size());
	return k.width;
}

// ---- detect + describe ----
struct orc_surf {
	DetectDescribeSurf* dd;
	SurfResult res;
};
void* orc_surf_create(int stable, const orc_fh_cfg* fh, const orc_surf_cfg* sd, const orc_ori_cfg* ori) {
	orc_surf* s = new orc_surf();
	s->dd = new DetectDescribeSurf(stable != 0, toFh(fh), toSd(sd), stable ? toSliding(ori) : ConfigSlidingIntegral(),
								   stable ? ConfigAverageIntegral() : toAverage(ori));
	return s;
}
void orc_surf_destroy(void* h) { orc_surf* s = (orc_surf*)h; delete s->dd; delete s; }
int orc_surf_detect(void* h, const orc_image* img, int threads) {
	orc_surf* s = (orc_surf*)h;
	s->dd->threads = threads;
	s->dd->detect(view(img), s->res);
	return (int)s->res.points.size();
}
// describe externally supplied points on the integral image of the last detect() (or of `img` when given)
int orc_surf_describe_points(void* h, const orc_image* img, const double* xys, int n, int threads) {
	orc_surf* s = (orc_surf*)h;
	s->dd->threads = threads;
	if (img) { s->dd->ii.reshape(img->width, img->height); integral_transform(view(img), s->dd->ii); }
	std::vector<ScalePoint> pts(n);
	for (int i = 0; i < n; i++) pts[i] = {xys[3 * i], xys[3 * i + 1], xys[3 * i + 2]};
	s->dd->describeAll(pts, s->res);
	return n;
}
// colour SURF on a planar image: bands[nb] share one shape; the descriptor has nb * dof values
int orc_surf_detect_planar(void* h, const orc_image* bands, int nb, int threads) {
	orc_surf* s = (orc_surf*)h;
	s->dd->threads = threads;
	std::vector<GrayF32> b(nb);
	for (int i = 0; i < nb; i++) b[i] = view(&bands[i]);
	s->dd->detectPlanar(b, s->res);
	return (int)s->res.points.size();
}
// DescribePointSurfPlanar.describe for caller-supplied (x, y, scale) on the planar image of the last orc_surf_detect_planar
int orc_surf_describe_points_planar(void* h, const double* xys, int n, int threads) {
	orc_surf* s = (orc_surf*)h;
	s->dd->threads = threads;
	std::vector<ScalePoint> pts(n);
	for (int i = 0; i < n; i++) pts[i] = {xys[3 * i], xys[3 * i + 1], xys[3 * i + 2]};
	s->dd->describeAllPlanar(pts, s->res);
	return n;
}
int orc_surf_detect_u8(void* h, const uint8_t* img, int start, int stride, int w, int hh, int threads) {
	orc_surf* s = (orc_surf*)h;
	s->dd->threads = threads;
	GrayU8v v{img, start, stride, w, hh};
	surfDetectU8(*s->dd, v, s->res);
	return (int)s->res.points.size();
}
void orc_surf_fetch(void* h, double* xys, double* angle, uint8_t* white, double* desc) {
	orc_surf* s = (orc_surf*)h;
	size_t n = s->res.points.size();
	for (size_t i = 0; i < n; i++) {
		if (xys) { xys[3 * i] = s->res.points[i].x; xys[3 * i + 1] = s->res.points[i].y; xys[3 * i + 2] = s->res.points[i].scale; }
		if (angle) angle[i] = s->res.angles[i];
		if (white) white[i] = s->res.white[i];
	}
	if (desc && n) std::memcpy(desc, s->res.desc.data(), sizeof(double) * s->res.desc.size());
}
void orc_surf_integral(void* h, float* out) {
	orc_surf* s = (orc_surf*)h;
	const GrayF32& ii = s->dd->ii;
	for (int y = 0; y < ii.height; y++) std::memcpy(out + (size_t)y * ii.width, ii.data + ii.startIndex + y * ii.stride, sizeof(float) * ii.width);
}

// ---- association ----
void orc_associate_l2(const double* src, int ns, const double* dst, int nd, int dof, double maxErr, int backwards, int* pairs, double* fit, int threads) {
	std::vector<double> work;
	associateGreedy(ns, nd, [&](int i, int j) { return euclideanSq(src + (size_t)i * dof, dst + (size_t)j * dof, dof); }, maxErr, backwards != 0, pairs, fit,
					work, threads);
}
// ScoreAssociateEuclidean_F64 = sqrt(euclideanSq) (DescriptorDistance.java:36-46); used by the reference's TestAssociateGreedy literals
void orc_associate_euclidean(const double* src, int ns, const double* dst, int nd, int dof, double maxErr, int backwards, int* pairs, double* fit, int threads) {
	std::vector<double> work;
	associateGreedy(ns, nd, [&](int i, int j) { return std::sqrt(euclideanSq(src + (size_t)i * dof, dst + (size_t)j * dof, dof)); }, maxErr, backwards != 0,
					pairs, fit, work, threads);
}
void orc_associate_hamming(const int32_t* src, int ns, const int32_t* dst, int nd, int words, double maxErr, int backwards, int* pairs, double* fit,
						   int threads) {
	std::vector<double> work;
	associateGreedy(ns, nd, [&](int i, int j) { return (double)hamming(src + (size_t)i * words, dst + (size_t)j * words, words); }, maxErr,
					backwards != 0, pairs, fit, work, threads);
}
double orc_euclidean_sq(const double* a, const double* b, int n) { return euclideanSq(a, b, n); }
int orc_hamming_word(int32_t v) { return hammingWord(v); }
int orc_hamming(const int32_t* a, const int32_t* b, int n) { return hamming(a, b, n); }

// ---- BRIEF ----
// definition: writes samplePoints (2*numPoints ints) and compare (2*numPoints ints)
void orc_brief_definition(int64_t seed, int radius, int numPoints, int* samplePoints, int* compare) {
	JavaRandom rand(seed);
	BinaryCompareDefinition def = briefGaussian2(rand, radius, numPoints);
	for (int i = 0; i < numPoints; i++) {
		samplePoints[2 * i] = def.samplePoints[i].x; samplePoints[2 * i + 1] = def.samplePoints[i].y;
		compare[2 * i] = def.compare[i].x; compare[2 * i + 1] = def.compare[i].y;
	}
}
void orc_brief_describe(const orc_image* img, int radius, int numPoints, const int* samplePoints, const int* compare, const double* xy, int n, int32_t* out) {
	BinaryCompareDefinition def; def.radius = radius;
	def.samplePoints.resize(numPoints); def.compare.resize(numPoints);
	for (int i = 0; i < numPoints; i++) {
		def.samplePoints[i] = {samplePoints[2 * i], samplePoints[2 * i + 1]};
		def.compare[i] = {compare[2 * i], compare[2 * i + 1]};
	}
	GrayF32 v = view(img);
	DescribeBinaryCompare_F32 d(def);
	d.setImage(v);
	int words = (numPoints + 31) / 32;
	for (int i = 0; i < n; i++) d.process((int)xy[2 * i], (int)xy[2 * i + 1], out + (size_t)i * words);
}

// ---- convolution / blur / gradient / pyramid ----
void orc_conv_h(const float* kernel, int kw, int koff, const orc_image* in, const orc_image* out, int threads) { GrayF32 o = view(out); convolveHorizontalNoBorder(kernel, kw, koff, view(in), o, threads); }
void orc_conv_v(const float* kernel, int kw, int koff, const orc_image* in, const orc_image* out, int threads) { GrayF32 o = view(out); convolveVerticalNoBorder(kernel, kw, koff, view(in), o, threads); }
void orc_conv_norm_h(const float* kernel, int kw, int koff, const orc_image* in, const orc_image* out, int threads) { GrayF32 o = view(out); convolveNormalizedHorizontal(kernel, kw, koff, view(in), o, threads); }
void orc_conv_norm_v(const float* kernel, int kw, int koff, const orc_image* in, const orc_image* out, int threads) { GrayF32 o = view(out); convolveNormalizedVertical(kernel, kw, koff, view(in), o, threads); }
void orc_gaussian_blur(const orc_image* in, const orc_image* out, double sigma, int radius, const orc_image* storage, int threads) {
	GrayF32 o = view(out), s = view(storage);
	blurGaussian(view(in), o, sigma, radius, s, threads);
}
void orc_sobel(const orc_image* in, const orc_image* dx, const orc_image* dy, int borderZero, int threads) {
	GrayF32 x = view(dx), y = view(dy);
	gradientSobel(view(in), x, y, borderZero != 0, threads);
}
void orc_three(const orc_image* in, const orc_image* dx, const orc_image* dy, int borderZero, int threads) {
	GrayF32 x = view(dx), y = view(dy);
	gradientThree(view(in), x, y, borderZero != 0, threads);
}
// down-sampling normalised convolution; returns 0, or -1 where the reference throws / reads out of the image
int orc_conv_down_norm(int vertical, const float* kernel, int kw, const orc_image* in, const orc_image* out, int skip) {
	try {
		GrayF32 o = view(out);
		if (vertical) convolveDownNormalizedVertical(kernel, kw, view(in), o, skip);
		else convolveDownNormalizedHorizontal(kernel, kw, view(in), o, skip);
	} catch (const DownConvError&) { return -1; }
	return 0;
}
int orc_down_max_side(int side, int skip, int radius) { return downComputeMaxSide(side, skip, radius); }
int orc_down_offset(int skip, int radius) { return downComputeOffset(skip, radius); }
// PyramidDiscreteSampleBlur.process: layers are written back to back into `out` (capacity outCap floats),
// dims[2*i], dims[2*i+1] = width, height of layer i, sigmas[i] = getSigma(i).  Returns floats written or -1.
long orc_pyramid(const float* kernel, int kw, double sigma, const int* scales, int n, const orc_image* in, float* out, long outCap, int* dims, double* sigmas) {
	try {
		PyramidDiscreteSampleBlur pyr(kernel, kw, sigma, scales, n);
		pyr.process(view(in));
		long off = 0;
		for (int i = 0; i < n; i++) {
			const GrayF32& l = pyr.layers[i];
			dims[2 * i] = l.width; dims[2 * i + 1] = l.height;
			sigmas[i] = pyr.sigmas[i];
			if (off + (long)l.width * l.height > outCap) return -1;
			for (int y = 0; y < l.height; y++)
				for (int x = 0; x < l.width; x++) out[off++] = l.get(x, y);
		}
		return off;
	} catch (const DownConvError&) { return -1; }
}
// GradientCornerIntensity.process; out must be a dense width x height image.  Returns -1 where the reference throws.
int orc_ssd_corner(const orc_image* dx, const orc_image* dy, int radius, int kind, float kappa, float* out) {
	try {
		GrayF32 inten(dx->width, dx->height);
		ssdCornerF32(view(dx), view(dy), radius, kind, kappa, inten);
		std::memcpy(out, inten.data, sizeof(float) * (size_t)dx->width * dx->height);
	} catch (const std::exception&) { return -1; }
	return 0;
}
void orc_conv2d(const float* kernel, int kw, int koff, const orc_image* in, const orc_image* out) { GrayF32 o = view(out); convolve2DNoBorder(kernel, kw, koff, view(in), o); }
int orc_blur_mean(const orc_image* in, const orc_image* out, int radiusX, int radiusY, const orc_image* storage) {
	try { GrayF32 o = view(out), st = view(storage); blurMean(view(in), o, radiusX, radiusY, st); } catch (const std::exception&) { return -1; }
	return 0;
}
int orc_blur_median(const orc_image* in, const orc_image* out, int radius) {
	try { GrayF32 o = view(out); blurMedian(view(in), o, radius); } catch (const std::exception&) { return -1; }
	return 0;
}
// ---- integer image variants, stage level ----
void orc_integral_u8(const uint8_t* in, int inStart, int inStride, int w, int h, int32_t* out, int outStart, int outStride) {
	GrayU8v i{in, inStart, inStride, w, h};
	GrayS32v o{out, outStart, outStride, w, h};
	integral_transform_u8(i, o);
}
void orc_hessian_s32(const int32_t* ii, int iiStart, int iiStride, int w, int h, int skip, int size, const orc_image* out) {
	GrayS32v v{(int32_t*)ii, iiStart, iiStride, w, h};
	GrayF32 o = view(out);
	hessian_s32(v, skip, size, o);
}
int orc_fh_detect_s32(const int32_t* ii, int iiStart, int iiStride, int w, int h, const orc_fh_cfg* cfg, double* out, int cap, int threads) {
	GrayS32v v{(int32_t*)ii, iiStart, iiStride, w, h};
	FastHessianFeatureDetector det(toFh(cfg));
	det.threads = threads;
	det.detect(v);
	int n = (int)det.foundPoints.size();
	for (int i = 0; i < n && i < cap; i++) { out[3 * i] = det.foundPoints[i].x; out[3 * i + 1] = det.foundPoints[i].y; out[3 * i + 2] = det.foundPoints[i].scale; }
	return n;
}
void orc_brief_u8(const uint8_t* img, int start, int stride, int w, int h, int radius, int numPoints, const int* samplePoints, const int* compare, const double* xy,
				  int n, int32_t* out) {
	GrayU8v v{img, start, stride, w, h};
	const int words = (numPoints + 31) / 32;
	for (int i = 0; i < n; i++) brief_u8(v, radius, numPoints, samplePoints, compare, (int)xy[2 * i], (int)xy[2 * i + 1], out + (size_t)i * words);
}
void orc_subsample(const orc_image* in, const orc_image* out, int skip) { GrayF32 o = view(out); pyramidSubsample(view(in), o, skip); }

}  // extern "C"
