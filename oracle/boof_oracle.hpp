// TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of BoofCV's detect -> describe -> associate path.
//
// This file is the checker for the HIP product in boofcv_amd/.  Nothing in the product may include,
// link or call it; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
//
// Every function follows the order of floating point operations of the Java reference
// (waicool20/BoofCV 0.35-SNAPSHOT) and cites the file:line it restates.  Path abbreviations:
//   F: = main/boofcv-feature/src/main/java/boofcv/     I: = main/boofcv-ip/src/main/java/boofcv/
//   T: = main/boofcv-types/src/main/java/boofcv/
//
// Parity status: there is no JVM in the build container, so the restatement is pinned by the
// reference's own known-answer tests (tests/test_oracle_*.py re-express them) -- not by running Java.
// Third-party pieces that are not in the reference tree (ddogleg UtilGaussian / QuickSort_F64 /
// QuickSelect, georegression UtilAngle, java.util.Random, java.lang.Math transcendental functions)
// are restated from their public contracts; where that contract leaves freedom (sort tie order,
// last-ulp of sin/cos/atan2/exp/log) the result is "parity unpinned" and documented in DESIGN.md.
//
// Build: g++ -O2 -std=c++17 -ffp-contract=off -fno-fast-math (see oracle/Makefile).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cfloat>
#include <vector>
#include <algorithm>
#include <stdexcept>
#include <functional>

namespace oracle {

// ------------------------------------------------------------------------------------------------
// java.util.Random (JDK class; behaviour fixed by the Java SE specification) -- SURVEY A.10
// ------------------------------------------------------------------------------------------------
struct JavaRandom {
	int64_t seed;
	bool haveNextNextGaussian = false;
	double nextNextGaussian = 0;

	explicit JavaRandom(int64_t s) { setSeed(s); }
	void setSeed(int64_t s) {
		seed = (s ^ 0x5DEECE66DLL) & ((1LL << 48) - 1);
		haveNextNextGaussian = false;
	}
	int32_t next(int bits) {
		seed = (int64_t)(((uint64_t)seed * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1));
		return (int32_t)(seed >> (48 - bits));
	}
	int32_t nextInt() { return next(32); }
	int32_t nextInt(int32_t bound) {
		if (bound <= 0) throw std::invalid_argument("bound must be positive");
		int32_t r = next(31);
		int32_t m = bound - 1;
		if ((bound & m) == 0) {
			r = (int32_t)(((int64_t)bound * (int64_t)r) >> 31);
		} else {
			// int overflow in (u - r + m) is the loop condition in Java; emulate with wrap-around
			for (int32_t u = r; (int32_t)((uint32_t)u - (uint32_t)(r = u % bound) + (uint32_t)m) < 0; u = next(31)) {}
		}
		return r;
	}
	int64_t nextLong() { return (int64_t)((uint64_t)(int64_t)next(32) << 32) + (int64_t)next(32); }
	bool nextBoolean() { return next(1) != 0; }
	float nextFloat() { return next(24) / (float)(1 << 24); }
	double nextDouble() { return (double)(((int64_t)next(26) << 27) + next(27)) * 0x1.0p-53; }
	double nextGaussian() {
		if (haveNextNextGaussian) {
			haveNextNextGaussian = false;
			return nextNextGaussian;
		}
		double v1, v2, s;
		do {
			v1 = 2 * nextDouble() - 1;
			v2 = 2 * nextDouble() - 1;
			s = v1 * v1 + v2 * v2;
		} while (s >= 1 || s == 0);
		// Java uses StrictMath (fdlibm) here; libm log may differ in the last ulp: parity unpinned
		double multiplier = std::sqrt(-2 * std::log(s) / s);
		nextNextGaussian = v2 * multiplier;
		haveNextNextGaussian = true;
		return v1 * multiplier;
	}
};

// ------------------------------------------------------------------------------------------------
// GrayF32 view: pixel (x,y) = data[startIndex + y*stride + x]   T:struct/image/ImageBase.java:34-52
// ------------------------------------------------------------------------------------------------
struct GrayF32 {
	float* data = nullptr;
	int startIndex = 0, stride = 0, width = 0, height = 0;
	std::vector<float> storage;  // optional owner

	GrayF32() {}
	GrayF32(int w, int h) { reshape(w, h); }
	GrayF32(float* d, int start, int str, int w, int h) : data(d), startIndex(start), stride(str), width(w), height(h) {}
	void reshape(int w, int h) {
		// ImageGray.reshape: grows storage if needed, stride=width, startIndex stays 0 for owned images
		if ((size_t)w * h > storage.size()) storage.resize((size_t)w * h);
		data = storage.data();
		startIndex = 0; stride = w; width = w; height = h;
	}
	inline float get(int x, int y) const { return data[startIndex + y * stride + x]; }
	inline void set(int x, int y, float v) { data[startIndex + y * stride + x] = v; }
	inline bool isInBounds(int x, int y) const { return x >= 0 && x < width && y >= 0 && y < height; }
};

// I:alg/misc/ImageMiscOps.java:2674-2685  fillUniform(GrayF32,Random,min,max)
inline void fillUniform(GrayF32& img, JavaRandom& rand, float min, float max) {
	float range = max - min;
	for (int y = 0; y < img.height; y++) {
		int index = img.startIndex + y * img.stride;
		for (int x = 0; x < img.width; x++) img.data[index++] = rand.nextFloat() * range + min;
	}
}
// I:alg/misc/ImageMiscOps.java:2721-2733  fillGaussian(GrayF32,Random,mean,sigma,lower,upper)
inline void fillGaussian(GrayF32& img, JavaRandom& rand, double mean, double sigma, float lower, float upper) {
	for (int y = 0; y < img.height; y++) {
		int index = img.startIndex + y * img.stride;
		for (int x = 0; x < img.width; x++) {
			float value = (float)(rand.nextGaussian() * sigma + mean);
			if (value < lower) value = lower;
			if (value > upper) value = upper;
			img.data[index++] = value;
		}
	}
}

// ------------------------------------------------------------------------------------------------
// Integral image   I:alg/transform/ii/impl/ImplIntegralImageOps.java
// ------------------------------------------------------------------------------------------------
// :42-66 transform(GrayF32,GrayF32)
inline void integral_transform(const GrayF32& input, GrayF32& transformed) {
	int indexSrc = input.startIndex;
	int indexDst = transformed.startIndex;
	int end = indexSrc + input.width;
	float total = 0;
	for (; indexSrc < end; indexSrc++) {
		total += input.data[indexSrc];
		transformed.data[indexDst++] = total;
	}
	for (int y = 1; y < input.height; y++) {
		indexSrc = input.startIndex + input.stride * y;
		indexDst = transformed.startIndex + transformed.stride * y;
		int indexPrev = indexDst - transformed.stride;
		end = indexSrc + input.width;
		total = 0;
		for (; indexSrc < end; indexSrc++) {
			total += input.data[indexSrc];
			transformed.data[indexDst++] = transformed.data[indexPrev++] + total;
		}
	}
}
// :185-193 block_unsafe
inline float block_unsafe(const GrayF32& ii, int x0, int y0, int x1, int y1) {
	float br = ii.data[ii.startIndex + y1 * ii.stride + x1];
	float tr = ii.data[ii.startIndex + y0 * ii.stride + x1];
	float bl = ii.data[ii.startIndex + y1 * ii.stride + x0];
	float tl = ii.data[ii.startIndex + y0 * ii.stride + x0];
	return br - tr - bl + tl;
}
// :195-214 block_zero
inline float block_zero(const GrayF32& ii, int x0, int y0, int x1, int y1) {
	x0 = std::min(x0, ii.width - 1);
	y0 = std::min(y0, ii.height - 1);
	x1 = std::min(x1, ii.width - 1);
	y1 = std::min(y1, ii.height - 1);
	float br = 0, tr = 0, bl = 0, tl = 0;
	if (x1 >= 0 && y1 >= 0) br = ii.data[ii.startIndex + y1 * ii.stride + x1];
	if (y0 >= 0 && x1 >= 0) tr = ii.data[ii.startIndex + y0 * ii.stride + x1];
	if (x0 >= 0 && y1 >= 0) bl = ii.data[ii.startIndex + y1 * ii.stride + x0];
	if (x0 >= 0 && y0 >= 0) tl = ii.data[ii.startIndex + y0 * ii.stride + x0];
	return br - tr - bl + tl;
}

// I:alg/transform/ii/IntegralKernel.java:30-48 ; rectangles are (x0,y0,x1,y1), lower bound exclusive
struct IntegralKernel {
	int n = 0;
	int x0[4], y0[4], x1[4], y1[4];
	int scales[4];
	void set(int i, int ax0, int ay0, int ax1, int ay1, int s) { x0[i] = ax0; y0[i] = ay0; x1[i] = ax1; y1[i] = ay1; scales[i] = s; }
};
// I:alg/transform/ii/DerivativeIntegralImage.java:102-119
inline IntegralKernel kernelDerivXX(int size) {
	IntegralKernel k; k.n = 2;
	int blockW = size / 3;
	int blockH = size - blockW - 1;
	int r1 = blockW / 2, r2 = blockW + r1, r3 = blockH / 2;
	k.set(0, -r2 - 1, -r3 - 1, r2, r3, 1);
	k.set(1, -r1 - 1, -r3 - 1, r1, r3, -3);
	return k;
}
// :121-137
inline IntegralKernel kernelDerivYY(int size) {
	IntegralKernel k; k.n = 2;
	int blockW = size / 3;
	int blockH = size - blockW - 1;
	int r1 = blockW / 2, r2 = blockW + r1, r3 = blockH / 2;
	k.set(0, -r3 - 1, -r2 - 1, r3, r2, 1);
	k.set(1, -r3 - 1, -r1 - 1, r3, r1, -3);
	return k;
}
// :139-158
inline IntegralKernel kernelDerivXY(int size) {
	IntegralKernel k; k.n = 4;
	int block = size / 3;
	k.set(0, -block - 1, -block - 1, -1, -1, 1);
	k.set(1, 0, -block - 1, block, -1, -1);
	k.set(2, 0, 0, block, block, 1);
	k.set(3, -block - 1, 0, -1, block, -1);
	return k;
}
// ImplIntegralImageOps.java:172-183 convolveSparse  (float * int scale: the int is widened to float)
inline float convolveSparse(const GrayF32& ii, const IntegralKernel& k, int x, int y) {
	float ret = 0;
	for (int i = 0; i < k.n; i++) {
		ret += block_zero(ii, x + k.x0[i], y + k.y0[i], x + k.x1[i], y + k.y1[i]) * (float)k.scales[i];
	}
	return ret;
}

// the same on a view whose 32-bit words are int32 (GrayS32 integral image): ImplIntegralImageOps.java:239-258 block_zero(GrayS32) + convolveSparse
inline int32_t convolveSparseInt(const GrayF32& iiView, const IntegralKernel& k, int x, int y) {
	const int32_t* d = reinterpret_cast<const int32_t*>(iiView.data);
	int32_t ret = 0;
	for (int i = 0; i < k.n; i++) {
		int x0 = std::min(x + k.x0[i], iiView.width - 1), y0 = std::min(y + k.y0[i], iiView.height - 1);
		int x1 = std::min(x + k.x1[i], iiView.width - 1), y1 = std::min(y + k.y1[i], iiView.height - 1);
		int32_t br = 0, tr = 0, bl = 0, tl = 0;
		if (x1 >= 0 && y1 >= 0) br = d[iiView.startIndex + y1 * iiView.stride + x1];
		if (y0 >= 0 && x1 >= 0) tr = d[iiView.startIndex + y0 * iiView.stride + x1];
		if (x0 >= 0 && y1 >= 0) bl = d[iiView.startIndex + y1 * iiView.stride + x0];
		if (x0 >= 0 && y0 >= 0) tl = d[iiView.startIndex + y0 * iiView.stride + x0];
		ret += (br - tr - bl + tl) * k.scales[i];
	}
	return ret;
}

// ------------------------------------------------------------------------------------------------
// Hessian-determinant intensity   F:alg/feature/detect/intensity/impl/ImplIntegralImageFeatureIntensity.java
// ------------------------------------------------------------------------------------------------
// :115-127 computeHessian
inline void computeHessian(const GrayF32& ii, GrayF32& intensity, const IntegralKernel& kerXX, const IntegralKernel& kerYY,
						   const IntegralKernel& kerXY, float norm, int y, int yy, int x, int xx) {
	float Dxx = convolveSparse(ii, kerXX, xx, yy);
	float Dyy = convolveSparse(ii, kerYY, xx, yy);
	float Dxy = convolveSparse(ii, kerXY, xx, yy);
	Dxx *= norm;
	Dxy *= norm;
	Dyy *= norm;
	float det = Dxx * Dyy - 0.81f * Dxy * Dxy;
	intensity.set(x, y, det);
}
// :45-67 hessianNaive
inline void hessianNaive(const GrayF32& ii, int skip, int size, GrayF32& intensity) {
	const int w = intensity.width, h = intensity.height;
	IntegralKernel kerXX = kernelDerivXX(size), kerYY = kernelDerivYY(size), kerXY = kernelDerivXY(size);
	float norm = 1.0f / (size * size);
	for (int y = 0; y < h; y++)
		for (int x = 0; x < w; x++) computeHessian(ii, intensity, kerXX, kerYY, kerXY, norm, y, y * skip, x, x * skip);
}
// :72-113 hessianBorder
inline void hessianBorder(const GrayF32& ii, int skip, int size, GrayF32& intensity) {
	const int w = intensity.width, h = intensity.height;
	IntegralKernel kerXX = kernelDerivXX(size), kerYY = kernelDerivYY(size), kerXY = kernelDerivXY(size);
	int radiusFeature = size / 2;
	const int borderOrig = radiusFeature + 1 + (skip - (radiusFeature + 1) % skip);
	const int border = borderOrig / skip;
	float norm = 1.0f / (size * size);
	for (int y = 0; y < h; y++) {
		int yy = y * skip;
		for (int x = 0; x < border; x++) computeHessian(ii, intensity, kerXX, kerYY, kerXY, norm, y, yy, x, x * skip);
		for (int x = w - border; x < w; x++) computeHessian(ii, intensity, kerXX, kerYY, kerXY, norm, y, yy, x, x * skip);
	}
	for (int x = border; x < w - border; x++) {
		int xx = x * skip;
		for (int y = 0; y < border; y++) computeHessian(ii, intensity, kerXX, kerYY, kerXY, norm, y, y * skip, x, xx);
		for (int y = h - border; y < h; y++) computeHessian(ii, intensity, kerXX, kerYY, kerXY, norm, y, y * skip, x, xx);
	}
}
// :132-213 hessianInner.  `threads`>1 mirrors ImplIntegralImageFeatureIntensity_MT (rows in parallel).
inline void hessianInner(const GrayF32& ii, int skip, int size, GrayF32& intensity, int threads = 1) {
	const int w = intensity.width, h = intensity.height;
	float norm = 1.0f / (size * size);
	int blockSmall = size / 3;
	int blockLarge = size - blockSmall - 1;
	int radiusFeature = size / 2;
	int radiusSkinny = blockLarge / 2;
	int blockW2 = 2 * blockSmall, blockW3 = 3 * blockSmall;
	int rowOff1 = blockSmall * ii.stride, rowOff2 = 2 * rowOff1, rowOff3 = 3 * rowOff1;
	const int borderOrig = radiusFeature + 1 + (skip - (radiusFeature + 1) % skip);
	const int border = borderOrig / skip;
	const int lostPixel = borderOrig - radiusFeature - 1;
	const int endY = h - border, endX = w - border;
	const float* d = ii.data;
	(void)threads;
#pragma omp parallel for num_threads(threads) if (threads > 1) schedule(static)
	for (int y = border; y < endY; y++) {
		int yy = y * skip;
		int indexDst = intensity.startIndex + y * intensity.stride + border;
		int indexTop = ii.startIndex + (yy - radiusSkinny - 1) * ii.stride + lostPixel;
		int indexBottom = indexTop + blockLarge * ii.stride;
		int indexL = ii.startIndex + (yy - radiusFeature - 1) * ii.stride + (radiusFeature - radiusSkinny) + lostPixel;
		int indexR = indexL + blockLarge;
		int indexY1 = ii.startIndex + (yy - blockSmall - 1) * ii.stride + (radiusFeature - blockSmall) + lostPixel;
		int indexY2 = indexY1 + blockSmall * ii.stride;
		int indexY3 = indexY2 + ii.stride;
		int indexY4 = indexY3 + blockSmall * ii.stride;
		for (int x = border; x < endX; x++, indexDst++) {
			float Dxx = d[indexBottom + blockW3] - d[indexTop + blockW3] - d[indexBottom] + d[indexTop];
			Dxx -= 3 * (d[indexBottom + blockW2] - d[indexTop + blockW2] - d[indexBottom + blockSmall] + d[indexTop + blockSmall]);
			float Dyy = d[indexR + rowOff3] - d[indexL + rowOff3] - d[indexR] + d[indexL];
			Dyy -= 3 * (d[indexR + rowOff2] - d[indexL + rowOff2] - d[indexR + rowOff1] + d[indexL + rowOff1]);
			int x3 = blockSmall + 1;
			int x4 = x3 + blockSmall;
			float Dxy = d[indexY2 + blockSmall] - d[indexY1 + blockSmall] - d[indexY2] + d[indexY1];
			Dxy -= d[indexY2 + x4] - d[indexY1 + x4] - d[indexY2 + x3] + d[indexY1 + x3];
			Dxy += d[indexY4 + x4] - d[indexY3 + x4] - d[indexY4 + x3] + d[indexY3 + x3];
			Dxy -= d[indexY4 + blockSmall] - d[indexY3 + blockSmall] - d[indexY4] + d[indexY3];
			Dxx *= norm;
			Dxy *= norm;
			Dyy *= norm;
			intensity.data[indexDst] = Dxx * Dyy - 0.81f * Dxy * Dxy;
			indexTop += skip; indexBottom += skip; indexL += skip; indexR += skip;
			indexY1 += skip; indexY2 += skip; indexY3 += skip; indexY4 += skip;
		}
	}
}
// F:alg/feature/detect/intensity/IntegralImageFeatureIntensity.java:43-56
inline void hessian(const GrayF32& ii, int skip, int size, GrayF32& intensity, int threads = 1) {
	hessianBorder(ii, skip, size, intensity);
	hessianInner(ii, skip, size, intensity, threads);
}

// ------------------------------------------------------------------------------------------------
// Non-maximum suppression   F:alg/feature/detect/extract/NonMaxBlock.java, NonMaxBlockSearchStrict.java
// ------------------------------------------------------------------------------------------------
struct Point2D_I16 { int16_t x, y; };
typedef std::vector<Point2D_I16> QueueCorner;

struct NonMaxBlockStrictMax {
	int radius = 1;           // ConfigExtract.radius
	float thresholdMax = 0;   // ConfigExtract.threshold
	int border = 0;           // ignoreBorder

	// NonMaxBlockSearchStrict.java:196-221 checkLocalMax
	void checkLocalMax(int x_c, int y_c, float peakVal, const GrayF32& img, QueueCorner& localMax) const {
		int x0 = x_c - radius, x1 = x_c + radius, y0 = y_c - radius, y1 = y_c + radius;
		if (x0 < 0) x0 = 0;
		if (y0 < 0) y0 = 0;
		if (x1 >= img.width) x1 = img.width - 1;
		if (y1 >= img.height) y1 = img.height - 1;
		for (int y = y0; y <= y1; y++) {
			int index = img.startIndex + y * img.stride + x0;
			for (int x = x0; x <= x1; x++) {
				float v = img.data[index++];
				if (v >= peakVal && !(x == x_c && y == y_c)) return;
			}
		}
		localMax.push_back({(int16_t)x_c, (int16_t)y_c});
	}
	// NonMaxBlockSearchStrict.java:56-79 Max.searchBlock
	void searchBlock(int x0, int y0, int x1, int y1, const GrayF32& img, QueueCorner& localMax) const {
		int peakX = 0, peakY = 0;
		float peakVal = -FLT_MAX;
		for (int y = y0; y < y1; y++) {
			int index = img.startIndex + y * img.stride + x0;
			for (int x = x0; x < x1; x++) {
				float v = img.data[index++];
				if (v > peakVal) { peakVal = v; peakX = x; peakY = y; }
			}
		}
		if (peakVal >= thresholdMax && peakVal != FLT_MAX) checkLocalMax(peakX, peakY, peakVal, img, localMax);
	}
	// NonMaxBlock.java:69-94 process ; output in block-raster order (USE_CONCURRENT=false order).
	// threads>1 mirrors NonMaxBlock_MT (block rows in parallel) but merges deterministically in row order.
	void process(const GrayF32& img, QueueCorner& localMax, int threads = 1) const {
		localMax.clear();
		int endX = img.width - border, endY = img.height - border;
		int step = radius + 1;
		if (threads <= 1) {
			for (int y = border; y < endY; y += step) {
				int y1 = y + step; if (y1 > endY) y1 = endY;
				for (int x = border; x < endX; x += step) {
					int x1 = x + step; if (x1 > endX) x1 = endX;
					searchBlock(x, y, x1, y1, img, localMax);
				}
			}
			return;
		}
		int nrows = endY > border ? (endY - border + step - 1) / step : 0;
		std::vector<QueueCorner> rows(nrows);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4)
		for (int r = 0; r < nrows; r++) {
			int y = border + r * step;
			int y1 = y + step; if (y1 > endY) y1 = endY;
			for (int x = border; x < endX; x += step) {
				int x1 = x + step; if (x1 > endX) x1 = endX;
				searchBlock(x, y, x1, y1, img, rows[r]);
			}
		}
		for (auto& r : rows) localMax.insert(localMax.end(), r.begin(), r.end());
	}
};

// F:alg/feature/detect/extract/NonMaxExtractorNaive.java:81-127 strictRule (set oracle for the block algorithm)
inline void nonmaxNaiveStrict(const GrayF32& intensity, int radius, float thresh, int border, QueueCorner& out) {
	out.clear();
	const int w = intensity.width, h = intensity.height;
	for (int y = border; y < h - border; y++) {
		for (int x = border; x < w - border; x++) {
			float val = intensity.get(x, y);
			if (val < thresh || val == FLT_MAX) continue;
			int x0 = std::max(0, x - radius), y0 = std::max(0, y - radius);
			int x1 = std::min(w, x + radius + 1), y1 = std::min(h, y + radius + 1);
			bool isMax = true;
			for (int i = y0; i < y1 && isMax; i++)
				for (int j = x0; j < x1; j++) {
					if (i == y && j == x) continue;
					if (val <= intensity.get(j, i)) { isMax = false; break; }
				}
			if (isMax) out.push_back({(int16_t)x, (int16_t)y});
		}
	}
}

// ------------------------------------------------------------------------------------------------
// Fast Hessian detector   F:alg/feature/detect/interest/FastHessianFeatureDetector.java
// ------------------------------------------------------------------------------------------------
struct ScalePoint { double x, y, scale; };

// F:abst/feature/detect/interest/ConfigFastHessian.java:33-70
struct ConfigFastHessian {
	float detectThreshold = 1;
	int extractRadius = 2;
	int maxFeaturesPerScale = -1;
	int initialSampleSize = 1;
	int initialSize = 9;
	int numberScalesPerOctave = 4;
	int numberOfOctaves = 4;
	int scaleStepSize = 6;
};

// :336-350 polyPeak(float)
inline float polyPeak(float lower, float middle, float upper) {
	float a = 0.5f * lower - middle + 0.5f * upper;
	float b = 0.5f * upper - 0.5f * lower;
	if (a == 0.0f) return 0.0f;
	return -b / (2.0f * a);
}
// T:struct/border/ImageBorderValue.java:121-142 Value_F32 with value 0 (getOutside returns the value)
inline float borderGet0(const GrayF32& img, int x, int y) { return img.isInBounds(x, y) ? img.get(x, y) : 0.0f; }
// :304-313 checkMax
inline bool checkMax(const GrayF32& inten, float bestScore, int c_x, int c_y) {
	for (int y = c_y - 1; y <= c_y + 1; y++)
		for (int x = c_x - 1; x <= c_x + 1; x++)
			if (borderGet0(inten, x, y) >= bestScore) return false;
	return true;
}

struct FastHessianFeatureDetector {
	ConfigFastHessian cfg;
	NonMaxBlockStrictMax extractor;  // FactoryInterestPointAlgs.java:171-183: ConfigExtract(extractRadius, detectThreshold, 0, true)
	GrayF32 intensity[3];
	int spaceIndex = 0;
	QueueCorner foundFeatures;
	std::vector<ScalePoint> foundPoints;
	int threads = 1;
	// optional capture of intermediate products for stage-level parity tests
	std::function<void(int octave, int level, int skip, int size, const GrayF32&)> onIntensity;
	std::function<void(int octave, int level, const QueueCorner&)> onNonMax;

	explicit FastHessianFeatureDetector(const ConfigFastHessian& c = ConfigFastHessian()) : cfg(c) {
		extractor.radius = c.extractRadius;
		extractor.thresholdMax = c.detectThreshold;
		extractor.border = 0;
	}

	// :156-188 detect (II = GrayF32, or the GrayS32 view of boof_oracle_int.hpp: only hessian() touches the pixels)
	template <class II>
	void detect(const II& integral) {
		if (intensity[0].storage.size() < (size_t)integral.width * integral.height)
			for (int i = 0; i < 3; i++) intensity[i].reshape(integral.width, integral.height);
		foundPoints.clear();
		std::vector<int> sizes(cfg.numberScalesPerOctave);
		int skip = cfg.initialSampleSize;
		int sizeStep = cfg.scaleStepSize;
		int octaveSize = cfg.initialSize;
		for (int octave = 0; octave < cfg.numberOfOctaves; octave++) {
			for (size_t i = 0; i < sizes.size(); i++) sizes[i] = octaveSize + (int)i * sizeStep;
			int maxSize = sizes[sizes.size() - 1];
			if (maxSize > integral.width || maxSize > integral.height) break;
			detectOctave(integral, skip, sizes, octave);
			skip += skip;
			octaveSize += sizeStep;
			sizeStep += sizeStep;
		}
	}
	// :198-221 detectOctave
	template <class II>
	void detectOctave(const II& integral, int skip, const std::vector<int>& featureSize, int octave) {
		int w = integral.width / skip, h = integral.height / skip;
		for (int i = 0; i < 3; i++) intensity[i].reshape(w, h);
		for (size_t i = 0; i < featureSize.size(); i++) {
			hessian(integral, skip, featureSize[i], intensity[spaceIndex], threads);
			if (onIntensity) onIntensity(octave, (int)i, skip, featureSize[i], intensity[spaceIndex]);
			spaceIndex++;
			if (spaceIndex >= 3) spaceIndex = 0;
			if (i >= 2) findLocalScaleSpaceMax(featureSize, (int)i - 1, skip, octave);
		}
	}
	// :230-298 findLocalScaleSpaceMax
	void findLocalScaleSpaceMax(const std::vector<int>& size, int level, int skip, int octave) {
		int index0 = spaceIndex, index1 = (spaceIndex + 1) % 3, index2 = (spaceIndex + 2) % 3;
		const GrayF32& inten0 = intensity[index0];
		const GrayF32& inten1 = intensity[index1];
		const GrayF32& inten2 = intensity[index2];

		extractor.border = size[level] / (2 * skip);
		extractor.process(inten1, foundFeatures, threads);
		if (onNonMax) onNonMax(octave, level, foundFeatures);

		int ignoreRadius = extractor.border + extractor.radius;
		int ignoreWidth = inten1.width - ignoreRadius;
		int ignoreHeight = inten1.height - ignoreRadius;

		int numberRemaining;
		QueueCorner selected;
		const QueueCorner* features;
		if (cfg.maxFeaturesPerScale > 0) {
			selectNBest(inten1, foundFeatures, cfg.maxFeaturesPerScale, selected);
			features = &selected;
			numberRemaining = cfg.maxFeaturesPerScale;
		} else {
			features = &foundFeatures;
			numberRemaining = INT32_MAX;
		}
		int levelSize = size[level];
		int sizeStep = levelSize - size[level - 1];

		for (size_t i = 0; i < features->size() && numberRemaining > 0; i++) {
			Point2D_I16 f = (*features)[i];
			if (f.x < ignoreRadius || f.x >= ignoreWidth || f.y < ignoreRadius || f.y >= ignoreHeight) continue;
			float val = inten1.get(f.x, f.y);
			if (checkMax(inten0, val, f.x, f.y) && checkMax(inten2, val, f.x, f.y)) {
				float peakX = polyPeak(inten1.get(f.x - 1, f.y), inten1.get(f.x, f.y), inten1.get(f.x + 1, f.y));
				float peakY = polyPeak(inten1.get(f.x, f.y - 1), inten1.get(f.x, f.y), inten1.get(f.x, f.y + 1));
				float peakS = polyPeak(borderGet0(inten0, f.x, f.y), inten1.get(f.x, f.y), borderGet0(inten2, f.x, f.y));
				float interpX = (f.x + peakX) * skip;
				float interpY = (f.y + peakY) * skip;
				float interpS = levelSize + peakS * sizeStep;
				double scale = 1.2 * interpS / 9.0;
				foundPoints.push_back({(double)interpX, (double)interpY, scale});
				numberRemaining--;
			}
		}
	}
	// org.ddogleg.sorting.QuickSelect.selectIndex(float[] data, int k, int maxIndex, int[] indexes) -- ddogleg is not in the reference
	// tree.  Its documentation names its source: the `select` routine of Numerical Recipes (3rd ed., 8.5) with an index array
	// carried through every exchange; that published routine is restated here.  Median-of-three with the median parked at l+1, Hoare
	// partition, iterate on the side holding k.  On return indexes[0..k) are the k smallest keys (in the partially sorted order the
	// exchanges leave them) and `data` is permuted the same way.  PARITY UNPINNED against the real ddogleg build: the only pin the
	// reference holds is TestSelectNBestFeatures.java:36-68 (the single best comes first for k = 3 of 4), which this order satisfies.
	static void quickSelectIndex(float* data, int k, int maxIndex, int* indexes) {
		int l = 0, ir = maxIndex - 1;
		for (int i = 0; i < maxIndex; i++) indexes[i] = i;
		auto swp = [&](int a, int b) { std::swap(data[a], data[b]); std::swap(indexes[a], indexes[b]); };
		for (;;) {
			if (ir <= l + 1) {
				if (ir == l + 1 && data[ir] < data[l]) swp(l, ir);
				return;
			}
			const int mid = (l + ir) >> 1, lp1 = l + 1;
			swp(mid, lp1);
			if (data[l] > data[ir]) swp(l, ir);
			if (data[lp1] > data[ir]) swp(lp1, ir);
			if (data[l] > data[lp1]) swp(l, lp1);
			int i = lp1, j = ir;
			const float a = data[lp1];
			const int indexA = indexes[lp1];
			for (;;) {
				do i++; while (data[i] < a);
				do j--; while (data[j] > a);
				if (j < i) break;
				swp(i, j);
			}
			data[lp1] = data[j]; data[j] = a;
			indexes[lp1] = indexes[j]; indexes[j] = indexA;
			if (j >= k) ir = j - 1;
			if (j <= k) l = i;
		}
	}
	// F:alg/feature/detect/extract/SelectNBestFeatures.java:51-93 (positive = true as FastHessianFeatureDetector.java:257 calls it;
	// positive = false keeps the smallest): keys are -intensity so the k smallest keys are the k largest intensities.
	static void selectNBest(const GrayF32& inten, const QueueCorner& orig, int target, QueueCorner& best, bool positive = true) {
		best.clear();
		if ((int)orig.size() <= target) { best = orig; return; }
		std::vector<int> idx(orig.size());
		std::vector<float> key(orig.size());
		for (size_t i = 0; i < idx.size(); i++) key[i] = positive ? -inten.get(orig[i].x, orig[i].y) : inten.get(orig[i].x, orig[i].y);
		quickSelectIndex(key.data(), target, (int)orig.size(), idx.data());
		for (int i = 0; i < target; i++) best.push_back(orig[idx[i]]);
	}
};

// ------------------------------------------------------------------------------------------------
// Gaussian kernels   I:factory/filter/kernel/FactoryKernelGaussian.java ; ddogleg UtilGaussian.computePDF
// ------------------------------------------------------------------------------------------------
// org.ddogleg.stats.UtilGaussian.computePDF(mean,sigma,sample) -- not in the reference tree; published formula
inline double computePDF(double mean, double sigma, double sample) {
	double delta = sample - mean;
	return std::exp(-delta * delta / (2.0 * sigma * sigma)) / (sigma * std::sqrt(2.0 * M_PI));
}
// :388 sigmaForRadius / :404 radiusForSigma
inline double sigmaForRadius(double radius, int order) {
	if (radius <= 0) throw std::invalid_argument("Radius must be > 0");
	return (radius * 2.0 + 1.0) / (5.0 + 0.8 * order);
}
inline int radiusForSigma(double sigma, int order) {
	if (sigma <= 0) throw std::invalid_argument("Sigma must be > 0");
	return (int)std::ceil((((5 + 0.8 * order) * sigma) - 1) / 2);
}
struct Kernel2D_F64 {
	int width = 0, offset = 0;
	std::vector<double> data;
	double get(int x, int y) const { return data[y * width + x]; }
	int getRadius() const { return width / 2; }
};
struct Kernel1D_F32 {
	int width = 0, offset = 0;
	std::vector<float> data;
};
// :240-260 gaussian1D_F64(sigma,radius,odd,normalize) (unnormalised part) + KernelMath.convolve2D(1D,1D) :365-383 +
// :297-306 gaussian2D_F64 + KernelMath.normalizeSumToOne :446-452
inline Kernel2D_F64 gaussian2D_F64(double sigma, int radius, bool odd, bool normalize) {
	std::vector<double> k1;
	if (odd) {
		for (int i = radius; i >= -radius; i--) k1.push_back(computePDF(0, sigma, i));
	} else {
		for (int i = radius; i > -radius; i--) k1.push_back(computePDF(0, sigma, i - 0.5));
	}
	Kernel2D_F64 ret;
	ret.width = (int)k1.size();
	ret.offset = ret.width / 2;
	ret.data.resize((size_t)ret.width * ret.width);
	int index = 0;
	for (int i = 0; i < ret.width; i++)
		for (int j = 0; j < ret.width; j++) ret.data[index++] = k1[i] * k1[j];
	if (normalize) {
		double total = 0;
		for (double v : ret.data) total += v;
		for (double& v : ret.data) v /= total;
	}
	return ret;
}
// :120-135 gaussian(DOF=2,isFloat=true,numBits=64,sigma,radius)
inline Kernel2D_F64 gaussian2D_F64_auto(double sigma, int radius) {
	if (radius <= 0) radius = radiusForSigma(sigma, 0);
	else if (sigma <= 0) sigma = sigmaForRadius(radius, 0);
	return gaussian2D_F64(sigma, radius, true, true);
}
// :418-448 gaussianWidth(sigma,width)
inline Kernel2D_F64 gaussianWidth(double sigma, int width) {
	if (sigma <= 0) sigma = sigmaForRadius(width / 2, 0);
	else if (width <= 0) throw std::invalid_argument("Must specify the width");
	if (width % 2 == 0) {
		int r = width / 2 - 1;
		Kernel2D_F64 ret;
		ret.width = width; ret.offset = width / 2;
		ret.data.resize((size_t)width * width);
		double sum = 0;
		for (int y = 0; y < width; y++) {
			double dy = y <= r ? std::abs(y - r) + 0.5 : std::abs(y - r - 1) + 0.5;
			for (int x = 0; x < width; x++) {
				double dx = x <= r ? std::abs(x - r) + 0.5 : std::abs(x - r - 1) + 0.5;
				double d = std::sqrt(dx * dx + dy * dy);
				double val = computePDF(0, sigma, d);
				ret.data[y * width + x] = val;
				sum += val;
			}
		}
		for (double& v : ret.data) v /= sum;
		return ret;
	}
	return gaussian2D_F64(sigma, width / 2, true, true);
}
// :218-238 gaussian1D_F32 + KernelMath.normalizeSumToOne(Kernel1D_F32) :417-423 ; :120-125 radius/sigma defaults
inline Kernel1D_F32 gaussian1D_F32(double sigma, int radius) {
	if (radius <= 0) radius = radiusForSigma(sigma, 0);
	else if (sigma <= 0) sigma = sigmaForRadius(radius, 0);
	Kernel1D_F32 ret;
	ret.width = radius * 2 + 1;
	ret.offset = ret.width / 2;
	for (int i = radius; i >= -radius; i--) ret.data.push_back((float)computePDF(0, sigma, i));
	float total = 0;
	for (float v : ret.data) total += v;
	for (float& v : ret.data) v /= total;
	return ret;
}

// ------------------------------------------------------------------------------------------------
// Sparse gradient from the integral image   I:alg/transform/ii/SparseIntegralGradient_NoBorder*.java
// ------------------------------------------------------------------------------------------------
struct SparseIntegralGradient_NoBorder_F32 {
	const GrayF32* input = nullptr;
	int r = 0, w = 0;
	int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
	// SparseIntegralGradient_NoBorder.java:42-47 + _F32.java:38-44
	void setWidth(double width) {
		r = ((int)(width + 0.5)) / 2;
		if (r <= 0) r = 1;
		w = r * 2 + 1;
		x0 = y0 = -r - 1;
		x1 = y1 = r;
	}
	// T:struct/sparse/SparseScaleGradient.java:48-50
	bool isInBounds(int x, int y) const { return x + x0 >= 0 && y + y0 >= 0 && x + x1 < input->width && y + y1 < input->height; }
	// GrayS32 integral image (of a GrayU8 frame): `input` is then a view whose 32-bit words are int32 and the taps are combined in integer
	// arithmetic (I:alg/transform/ii/impl/SparseIntegralGradient_NoBorder_I32.java:46-76).  GradientValue_I32.getX() hands the consumer a
	// double; here the value travels as a float, which is exact below 2^24 (checked) -- 255 * (2r+1) * r stays below that for r < 180.
	bool intPixels = false;
	void computeInt(int x, int y, float& gx, float& gy) const {
		const GrayF32& in = *input;
		const int32_t* d = reinterpret_cast<const int32_t*>(in.data);
		int horizontalOffset = x - r - 1;
		int indexSrc1 = in.startIndex + (y - r - 1) * in.stride + horizontalOffset;
		int indexSrc2 = indexSrc1 + r * in.stride;
		int indexSrc3 = indexSrc2 + in.stride;
		int indexSrc4 = indexSrc3 + r * in.stride;
		int32_t p0 = d[indexSrc1], p1 = d[indexSrc1 + r], p2 = d[indexSrc1 + r + 1], p3 = d[indexSrc1 + w];
		int32_t p11 = d[indexSrc2], p4 = d[indexSrc2 + w];
		int32_t p10 = d[indexSrc3], p5 = d[indexSrc3 + w];
		int32_t p9 = d[indexSrc4], p8 = d[indexSrc4 + r], p7 = d[indexSrc4 + r + 1], p6 = d[indexSrc4 + w];
		int32_t left = p8 - p9 - p1 + p0;
		int32_t right = p6 - p7 - p3 + p2;
		int32_t top = p4 - p11 - p3 + p0;
		int32_t bottom = p6 - p9 - p5 + p10;
		int32_t ix = right - left, iy = bottom - top;
		if (std::abs((long long)ix) >= (1 << 24) || std::abs((long long)iy) >= (1 << 24)) throw std::runtime_error("integer gradient does not fit a float exactly");
		gx = (float)ix;
		gy = (float)iy;
	}
	// _F32.java:46-76
	void compute(int x, int y, float& gx, float& gy) const {
		if (intPixels) { computeInt(x, y, gx, gy); return; }
		const GrayF32& in = *input;
		int horizontalOffset = x - r - 1;
		int indexSrc1 = in.startIndex + (y - r - 1) * in.stride + horizontalOffset;
		int indexSrc2 = indexSrc1 + r * in.stride;
		int indexSrc3 = indexSrc2 + in.stride;
		int indexSrc4 = indexSrc3 + r * in.stride;
		const float* d = in.data;
		float p0 = d[indexSrc1], p1 = d[indexSrc1 + r], p2 = d[indexSrc1 + r + 1], p3 = d[indexSrc1 + w];
		float p11 = d[indexSrc2], p4 = d[indexSrc2 + w];
		float p10 = d[indexSrc3], p5 = d[indexSrc3 + w];
		float p9 = d[indexSrc4], p8 = d[indexSrc4 + r], p7 = d[indexSrc4 + r + 1], p6 = d[indexSrc4 + w];
		float left = p8 - p9 - p1 + p0;
		float right = p6 - p7 - p3 + p2;
		float top = p4 - p11 - p3 + p0;
		float bottom = p6 - p9 - p5 + p10;
		gx = right - left;
		gy = bottom - top;
	}
	// T:struct/sparse/SparseGradientSafe.java:56-61
	void computeSafe(int x, int y, float& gx, float& gy) const {
		if (isInBounds(x, y)) compute(x, y, gx, gy);
		else { gx = 0; gy = 0; }
	}
};

// georegression.metric.UtilAngle.dist(a,b) -- not in the reference tree: circular distance in [0,pi]
inline double utilAngleDist(double angA, double angB) {
	double diff = angA - angB;
	if (diff > M_PI) diff = diff - 2.0 * M_PI;
	else if (diff < -M_PI) diff = 2.0 * M_PI + diff;
	return std::abs(diff);
}

// ------------------------------------------------------------------------------------------------
// Orientation   F:alg/feature/orientation/OrientationIntegralBase.java + impl/*
// ------------------------------------------------------------------------------------------------
// F:abst/feature/orientation/ConfigSlidingIntegral.java:34-54
struct ConfigSlidingIntegral {
	double objectRadiusToScale = 1.0 / 2.0;  // 1/BoofDefaults.SURF_SCALE_TO_RADIUS
	double samplePeriod = 0.65;
	double windowSize = M_PI / 3.0;
	int radius = 8;
	double weightSigma = -1;
	int sampleWidth = 6;
};
// F:abst/feature/orientation/ConfigAverageIntegral.java:34-51
struct ConfigAverageIntegral {
	double objectRadiusToScale = 1.0 / 2.0;
	int radius = 6;
	double samplePeriod = 1;
	int sampleWidth = 6;
	double weightSigma = -1;
};

struct OrientationIntegralBase {
	const GrayF32* ii = nullptr;
	double scale = 1;
	int sampleRadius, sampleWidth;
	bool hasWeights = false;
	Kernel2D_F64 weights;
	int kernelWidth;
	double period;
	double objectRadiusToScale;
	SparseIntegralGradient_NoBorder_F32 g;

	// OrientationIntegralBase.java:75-92
	OrientationIntegralBase(double objectRadiusToScale_, int sampleRadius_, double period_, int kernelWidth_, double weightSigma) {
		objectRadiusToScale = objectRadiusToScale_;
		sampleRadius = sampleRadius_;
		period = period_;
		kernelWidth = kernelWidth_;
		sampleWidth = sampleRadius * 2 + 1;
		if (weightSigma != 0) {
			hasWeights = true;
			weights = gaussian2D_F64_auto(weightSigma, sampleRadius);
		}
		setObjectRadius(1.0 / objectRadiusToScale);
	}
	// :94-98
	void setObjectRadius(double radius) {
		scale = radius * objectRadiusToScale;
		g.setWidth(scale * kernelWidth);
	}
	void setImage(const GrayF32& integral) { ii = &integral; g.input = &integral; }
	void setIntPixels(bool v) { g.intPixels = v; }
};

// impl/ImplOrientationSlidingWindowIntegral.java
struct OrientationSlidingWindow : OrientationIntegralBase {
	double windowSize;
	std::vector<double> derivX, derivY, angles;
	std::vector<int> order;
	int total = 0;

	explicit OrientationSlidingWindow(const ConfigSlidingIntegral& c = ConfigSlidingIntegral())
		: OrientationIntegralBase(c.objectRadiusToScale, c.radius, c.samplePeriod, c.sampleWidth, c.weightSigma), windowSize(c.windowSize) {
		size_t n = (size_t)sampleWidth * sampleWidth;
		derivX.resize(n); derivY.resize(n); angles.resize(n); order.resize(n);
	}
	// :81-107 compute
	double compute(double c_x, double c_y) {
		double period_ = scale * this->period;
		double tl_x = c_x - sampleRadius * period_;
		double tl_y = c_y - sampleRadius * period_;
		computeGradient(tl_x, tl_y, period_);
		if (hasWeights) {
			for (int i = 0; i < total; i++) {
				double w = weights.data[i];
				derivX[i] *= w;
				derivY[i] *= w;
			}
		}
		for (int i = 0; i < total; i++) angles[i] = std::atan2(derivY[i], derivX[i]);
		// ddogleg QuickSort_F64.sort(angles,0,n,order): arg-sort ascending, data untouched, tie order
		// unspecified.  Oracle choice (SURVEY hard part 5): stable by (angle, index).
		for (size_t i = 0; i < order.size(); i++) order[i] = (int)i;
		std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return angles[a] < angles[b]; });
		return estimateAngle();
	}
	// :109-137 computeGradient
	void computeGradient(double tl_x, double tl_y, double samplePeriod) {
		tl_x += 0.5;
		tl_y += 0.5;
		total = 0;
		for (int y = 0; y < sampleWidth; y++) {
			for (int x = 0; x < sampleWidth; x++, total++) {
				int xx = (int)(tl_x + x * samplePeriod);
				int yy = (int)(tl_y + y * samplePeriod);
				if (g.isInBounds(xx, yy)) {
					float gx, gy;
					g.compute(xx, yy, gx, gy);
					derivX[total] = gx;
					derivY[total] = gy;
				} else {
					derivX[total] = 0;
					derivY[total] = 0;
				}
			}
		}
	}
	// :139-188 estimateAngle
	double estimateAngle() {
		int start = 0, end = 1;
		int startIndex = order[start];
		int endIndex = order[end];
		double sumX = derivX[startIndex], sumY = derivY[startIndex];
		double best = sumX * sumX + sumY * sumY;
		double bestX = sumX, bestY = sumY;
		double endAngle = angles[endIndex];
		while (start != total) {
			startIndex = order[start];
			double startAngle = angles[startIndex];
			while (utilAngleDist(startAngle, endAngle) <= windowSize) {
				sumX += derivX[endIndex];
				sumY += derivY[endIndex];
				double mag = sumX * sumX + sumY * sumY;
				if (mag > best) { best = mag; bestX = sumX; bestY = sumY; }
				end++;
				if (end >= total) end = 0;
				endIndex = order[end];
				endAngle = angles[endIndex];
				if (endIndex == startIndex) break;
			}
			sumX -= derivX[startIndex];
			sumY -= derivY[startIndex];
			start++;
		}
		return std::atan2(bestY, bestX);
	}
};

// F:alg/feature/describe/SurfDescribeOps.java:183-204 isInside(width,height,tl_x,tl_y,regionSize,sampleSize)
inline bool surfIsInsideRegion(int width, int height, double tl_x, double tl_y, double regionSize, double sampleSize) {
	int w = (int)(sampleSize + 0.5);
	int r = w / 2 + w % 2;
	int x0 = (int)(tl_x + 0.5) - r - 1;
	int y0 = (int)(tl_y + 0.5) - r - 1;
	if (x0 < 0 || y0 < 0) return false;
	int x1 = (int)(tl_x + regionSize + 0.5) + r;
	int y1 = (int)(tl_y + regionSize + 0.5) + r;
	if (x1 >= width || y1 >= height) return false;
	return true;
}

// impl/ImplOrientationAverageGradientIntegral.java:54-127
struct OrientationAverage : OrientationIntegralBase {
	explicit OrientationAverage(const ConfigAverageIntegral& c = ConfigAverageIntegral())
		: OrientationIntegralBase(c.objectRadiusToScale, c.radius, c.samplePeriod, c.sampleWidth, c.weightSigma) {}
	double compute(double c_x, double c_y) {
		double period_ = scale * this->period;
		double tl_x = c_x - sampleRadius * period_;
		double tl_y = c_y - sampleRadius * period_;
		bool safe = !surfIsInsideRegion(ii->width, ii->height, tl_x, tl_y, sampleWidth * period_, kernelWidth * scale);
		tl_x += 0.5;
		tl_y += 0.5;
		double Dx = 0, Dy = 0;
		int i = 0;
		for (int y = 0; y < sampleWidth; y++) {
			int pixelsY = (int)(tl_y + y * period_);
			for (int x = 0; x < sampleWidth; x++, i++) {
				int pixelsX = (int)(tl_x + x * period_);
				float gx, gy;
				if (safe) g.computeSafe(pixelsX, pixelsY, gx, gy);
				else g.compute(pixelsX, pixelsY, gx, gy);
				if (hasWeights) {
					double w = weights.data[i];
					Dx += w * gx;
					Dy += w * gy;
				} else {
					Dx += gx;
					Dy += gy;
				}
			}
		}
		return std::atan2(Dy, Dx);
	}
};

// ------------------------------------------------------------------------------------------------
// SURF descriptors   F:alg/feature/describe/DescribePointSurf.java, DescribePointSurfMod.java, SurfDescribeOps.java
// ------------------------------------------------------------------------------------------------
// java.lang.Math.round(double): floor(x + 1/2) with ties toward +inf
inline int64_t javaRound(double x) { return (int64_t)std::floor(x + 0.5); }

// SurfDescribeOps.java:120-159 isInside(ii,X,Y,radiusRegions,kernelSize,scale,c,s)
inline bool surfIsInside(const GrayF32& ii, double X, double Y, int radiusRegions, int kernelSize, double scale, double c, double s) {
	int c_x = (int)javaRound(X);
	int c_y = (int)javaRound(Y);
	kernelSize = (int)std::ceil(kernelSize * scale);
	int kernelRadius = kernelSize / 2 + (kernelSize % 2);
	int radius = (int)std::ceil(radiusRegions * scale);
	int kernelPaddingMinus = radius + kernelRadius + 1;
	int kernelPaddingPlus = radius + kernelRadius;
	if (c != 0 || s != 0) {
		double xx = std::abs(c * kernelPaddingMinus - s * kernelPaddingMinus);
		double yy = std::abs(s * kernelPaddingMinus + c * kernelPaddingMinus);
		double delta = xx > yy ? xx - kernelPaddingMinus : yy - kernelPaddingMinus;
		kernelPaddingMinus += (int)std::ceil(delta);
		kernelPaddingPlus += (int)std::ceil(delta);
	}
	int x0 = c_x - kernelPaddingMinus;
	if (x0 < 0) return false;
	int x1 = c_x + kernelPaddingPlus;
	if (x1 >= ii.width) return false;
	int y0 = c_y - kernelPaddingMinus;
	if (y0 < 0) return false;
	int y1 = c_y + kernelPaddingPlus;
	if (y1 >= ii.height) return false;
	return true;
}

// F:abst/feature/describe/ConfigSurfDescribe.java:34-78
struct ConfigSurfDescribe {
	int widthLargeGrid = 4;
	int widthSubRegion = 5;
	int widthSample = 3;
	bool useHaar = false;  // haar variant is out of scope; must stay false
	// Speed
	double weightSigma = 4.5;
	// Stability
	int overLap = 2;
	double sigmaLargeGrid = 2.5;
	double sigmaSubRegion = 2.5;
};

struct BrightFeature {
	std::vector<double> value;
	bool white = false;
};

// F:alg/descriptor/UtilFeature.java:101-114
inline void normalizeL2(double* v, int n) {
	double norm = 0;
	for (int i = 0; i < n; i++) norm += v[i] * v[i];
	if (norm == 0) return;
	norm = std::sqrt(norm);
	for (int i = 0; i < n; i++) v[i] /= norm;
}

struct DescribePointSurf {
	int widthLargeGrid, widthSubRegion, widthSample;
	double weightSigma;
	int featureDOF;
	const GrayF32* ii = nullptr;
	Kernel2D_F64 weight;
	SparseIntegralGradient_NoBorder_F32 gradient;
	int radiusDescriptor;

	// DescribePointSurf.java:110-141
	DescribePointSurf(int widthLargeGrid_, int widthSubRegion_, int widthSample_, double weightSigma_)
		: widthLargeGrid(widthLargeGrid_), widthSubRegion(widthSubRegion_), widthSample(widthSample_), weightSigma(weightSigma_) {
		int radius = (widthLargeGrid * widthSubRegion) / 2;
		weight = gaussianWidth(weightSigma, radius * 2);
		double div = weight.get(radius, radius);
		for (double& v : weight.data) v /= div;
		featureDOF = widthLargeGrid * widthLargeGrid * 4;
		radiusDescriptor = (widthLargeGrid * widthSubRegion) / 2;
	}
	explicit DescribePointSurf(const ConfigSurfDescribe& c) : DescribePointSurf(c.widthLargeGrid, c.widthSubRegion, c.widthSample, c.weightSigma) {}
	virtual ~DescribePointSurf() {}

	void setImage(const GrayF32& integral) { ii = &integral; gradient.input = &integral; }
	bool intPixels = false;   // see SparseIntegralGradient::intPixels
	void setIntPixels(bool v) { intPixels = v; gradient.intPixels = v; }

	// :169-179 describe(x,y,angle,scale,BrightFeature)
	void describe(double x, double y, double angle, double scale, BrightFeature& ret) {
		ret.value.resize(featureDOF);
		describeTuple(x, y, angle, scale, ret.value.data());
		normalizeL2(ret.value.data(), featureDOF);
		ret.white = computeLaplaceSign((int)(x + 0.5), (int)(y + 0.5), scale);
	}
	// :190-213 describe(x,y,angle,scale,TupleDesc_F64)
	void describeTuple(double x, double y, double angle, double scale, double* value) {
		double c = std::cos(angle), s = std::sin(angle);
		bool isInBounds = surfIsInside(*ii, x, y, radiusDescriptor, widthSample, scale, c, s);
		gradient.input = ii;
		gradient.setWidth(widthSample * scale);
		features(x, y, c, s, scale, !isInBounds, value);
	}
	inline void sample(bool safe, int px, int py, float& gx, float& gy) const {
		if (safe) gradient.computeSafe(px, py, gx, gy);
		else gradient.compute(px, py, gx, gy);
	}
	// :235-295 features
	virtual void features(double c_x, double c_y, double c, double s, double scale, bool safe, double* features) {
		int regionSize = widthLargeGrid * widthSubRegion;
		int regionR = regionSize / 2;
		int regionEnd = regionSize - regionR;
		int regionIndex = 0;
		c_x += 0.5;
		c_y += 0.5;
		for (int rY = -regionR; rY < regionEnd; rY += widthSubRegion) {
			for (int rX = -regionR; rX < regionEnd; rX += widthSubRegion) {
				double sum_dx = 0, sum_dy = 0, sum_adx = 0, sum_ady = 0;
				for (int i = 0; i < widthSubRegion; i++) {
					double regionY = (rY + i) * scale;
					for (int j = 0; j < widthSubRegion; j++) {
						double w = weight.get(regionR + rX + j, regionR + rY + i);
						double regionX = (rX + j) * scale;
						int pixelX = (int)(c_x + c * regionX - s * regionY);
						int pixelY = (int)(c_y + s * regionX + c * regionY);
						float gx, gy;
						sample(safe, pixelX, pixelY, gx, gy);
						double dx = w * gx;
						double dy = w * gy;
						double pdx = c * dx + s * dy;
						double pdy = -s * dx + c * dy;
						sum_dx += pdx;
						sum_adx += std::abs(pdx);
						sum_dy += pdy;
						sum_ady += std::abs(pdy);
					}
				}
				features[regionIndex++] = sum_dx;
				features[regionIndex++] = sum_adx;
				features[regionIndex++] = sum_dy;
				features[regionIndex++] = sum_ady;
			}
		}
	}
	// :305-313 computeLaplaceSign
	bool computeLaplaceSign(int x, int y, double scale) const {
		int s = (int)std::ceil(scale);
		IntegralKernel kerXX = kernelDerivXX(9 * s);
		IntegralKernel kerYY = kernelDerivYY(9 * s);
		if (intPixels) {
			// GIntegralImageOps.convolveSparse(GrayS32) returns an int that is widened to double
			double lapI = (double)convolveSparseInt(*ii, kerXX, x, y);
			lapI += (double)convolveSparseInt(*ii, kerYY, x, y);
			return lapI > 0;
		}
		double lap = convolveSparse(*ii, kerXX, x, y);
		lap += convolveSparse(*ii, kerYY, x, y);
		return lap > 0;
	}
};

// DescribePointSurfMod.java
struct DescribePointSurfMod : DescribePointSurf {
	int overLap;
	Kernel2D_F64 weightGrid, weightSub;
	std::vector<double> samplesX, samplesY;

	// :76-106 (super(...,weightSigma=1,...); radiusDescriptor overwritten)
	DescribePointSurfMod(int widthLargeGrid_, int widthSubRegion_, int widthSample_, int overLap_, double sigmaLargeGrid, double sigmaSubRegion)
		: DescribePointSurf(widthLargeGrid_, widthSubRegion_, widthSample_, 1), overLap(overLap_) {
		weightGrid = gaussianWidth(sigmaLargeGrid, widthLargeGrid);
		weightSub = gaussianWidth(sigmaSubRegion, widthSubRegion + 2 * overLap);
		double div = weightGrid.get(weightGrid.getRadius(), weightGrid.getRadius());
		for (double& v : weightGrid.data) v /= div;
		div = weightSub.get(weightSub.getRadius(), weightSub.getRadius());
		for (double& v : weightSub.data) v /= div;
		int sampleWidth = widthLargeGrid * widthSubRegion + overLap * 2;
		samplesX.resize((size_t)sampleWidth * sampleWidth);
		samplesY.resize((size_t)sampleWidth * sampleWidth);
		radiusDescriptor = (widthLargeGrid * widthSubRegion) / 2 + overLap;
	}
	explicit DescribePointSurfMod(const ConfigSurfDescribe& c)
		: DescribePointSurfMod(c.widthLargeGrid, c.widthSubRegion, c.widthSample, c.overLap, c.sigmaLargeGrid, c.sigmaSubRegion) {}

	// :121-192 features
	void features(double c_x, double c_y, double c, double s, double scale, bool safe, double* features) override {
		int regionSize = widthLargeGrid * widthSubRegion;
		int totalSampleWidth = widthSubRegion + overLap * 2;
		int regionR = regionSize / 2;
		int regionEnd = regionSize - regionR;
		int sampleGridWidth = regionSize + 2 * overLap;
		int regionIndex = 0;
		c_x += 0.5;
		c_y += 0.5;
		int index = 0;
		for (int rY = -regionR - overLap; rY < regionEnd + overLap; rY++) {
			double regionY = rY * scale;
			for (int rX = -regionR - overLap; rX < regionEnd + overLap; rX++, index++) {
				double regionX = rX * scale;
				int pixelX = (int)(c_x + c * regionX - s * regionY);
				int pixelY = (int)(c_y + s * regionX + c * regionY);
				float gx, gy;
				sample(safe, pixelX, pixelY, gx, gy);
				samplesX[index] = gx;
				samplesY[index] = gy;
			}
		}
		int indexGridWeight = 0;
		for (int rY = -regionR; rY < regionEnd; rY += widthSubRegion) {
			for (int rX = -regionR; rX < regionEnd; rX += widthSubRegion) {
				double sum_dx = 0, sum_dy = 0, sum_adx = 0, sum_ady = 0;
				for (int i = 0; i < totalSampleWidth; i++) {
					index = (rY + regionR + i) * sampleGridWidth + rX + regionR;
					for (int j = 0; j < totalSampleWidth; j++, index++) {
						double w = weightSub.get(j, i);
						double dx = w * samplesX[index];
						double dy = w * samplesY[index];
						double pdx = c * dx + s * dy;
						double pdy = -s * dx + c * dy;
						sum_dx += pdx;
						sum_adx += std::abs(pdx);
						sum_dy += pdy;
						sum_ady += std::abs(pdy);
					}
				}
				double w = weightGrid.data[indexGridWeight++];
				features[regionIndex++] = w * sum_dx;
				features[regionIndex++] = w * sum_adx;
				features[regionIndex++] = w * sum_dy;
				features[regionIndex++] = w * sum_ady;
			}
		}
	}
};

// ------------------------------------------------------------------------------------------------
// detect + describe   F:abst/feature/detdesc/WrapDetectDescribeSurf.java ; FactoryDetectDescribe.java:118-135,209-226
// ------------------------------------------------------------------------------------------------
struct SurfResult {
	std::vector<ScalePoint> points;
	std::vector<double> angles;
	std::vector<double> desc;     // n*64
	std::vector<uint8_t> white;   // n
};

struct DetectDescribeSurf {
	bool stable;
	FastHessianFeatureDetector detector;
	OrientationSlidingWindow oriSliding;
	OrientationAverage oriAverage;
	DescribePointSurfMod describeMod;
	DescribePointSurf describeFast;
	GrayF32 ii;
	bool intPixels = false;   // ii holds int32 words: the GrayS32 integral image of a GrayU8 frame (detectU8 in boof_oracle_int.hpp)
	int threads = 1;

	DetectDescribeSurf(bool stable_, const ConfigFastHessian& fh = ConfigFastHessian(), const ConfigSurfDescribe& sd = ConfigSurfDescribe(),
					   const ConfigSlidingIntegral& so = ConfigSlidingIntegral(), const ConfigAverageIntegral& ao = ConfigAverageIntegral())
		: stable(stable_), detector(fh), oriSliding(so), oriAverage(ao), describeMod(sd), describeFast(sd) {}

	int dof() const { return stable ? describeMod.featureDOF : describeFast.featureDOF; }

	// WrapDetectDescribeSurf.java:93-128 (threads>1: WrapDetectDescribeSurf_MT.java:45-61, keypoint blocks in parallel)
	void detect(const GrayF32& input, SurfResult& out) {
		intPixels = false;
		ii.reshape(input.width, input.height);
		integral_transform(input, ii);
		detector.threads = threads;
		detector.detect(ii);
		describeAll(detector.foundPoints, out);
	}
	// ---- colour SURF: SurfPlanar_to_DetectDescribePoint.detect (F:abst/feature/detdesc/SurfPlanar_to_DetectDescribePoint.java:62-77) +
	// DetectDescribeSurfPlanar.detect/describe (F:alg/feature/detdesc/DetectDescribeSurfPlanar.java:91-124) +
	// DescribePointSurfPlanar.describe (F:alg/feature/describe/DescribePointSurfPlanar.java:100-114).
	// Differences from the grey wrapper, kept as in the reference: the orientation's object radius is p.scale (not 2*scale), the bands'
	// descriptors are concatenated UN-normalised and normalised once as a whole, the Laplacian sign comes from the grey integral image.
	std::vector<GrayF32> bandII;
	GrayF32 grayAvg;
	// ImplConvertPlanarToGray.average(Planar<GrayF32>) (I:core/image/impl/ImplConvertPlanarToGray.java:296-336)
	static void averagePlanar(const std::vector<GrayF32>& bands, GrayF32& to) {
		const int nb = (int)bands.size();
		to.reshape(bands[0].width, bands[0].height);
		for (int y = 0; y < to.height; y++)
			for (int x = 0; x < to.width; x++) {
				if (nb == 1) { to.set(x, y, bands[0].get(x, y)); continue; }
				float sum;
				if (nb == 3) { sum = bands[0].get(x, y); sum += bands[1].get(x, y); sum += bands[2].get(x, y); to.set(x, y, sum / 3); }
				else { sum = 0; for (int b = 0; b < nb; b++) sum += bands[b].get(x, y); to.set(x, y, sum / nb); }
			}
	}
	void detectPlanar(const std::vector<GrayF32>& bands, SurfResult& out) {
		averagePlanar(bands, grayAvg);
		ii.reshape(grayAvg.width, grayAvg.height);
		integral_transform(grayAvg, ii);
		bandII.resize(bands.size());
		for (size_t b = 0; b < bands.size(); b++) { bandII[b].reshape(bands[b].width, bands[b].height); integral_transform(bands[b], bandII[b]); }
		detector.threads = threads;
		detector.detect(ii);
		describeAllPlanar(detector.foundPoints, out);
	}
	void describeAllPlanar(const std::vector<ScalePoint>& pts, SurfResult& out) {
		const int n = (int)pts.size(), D = dof(), nb = (int)bandII.size();
		out.points = pts;
		out.angles.assign(n, 0);
		out.desc.assign((size_t)n * D * nb, 0);
		out.white.assign(n, 0);
#pragma omp parallel num_threads(threads) if (threads > 1)
		{
			OrientationSlidingWindow os = oriSliding;
			OrientationAverage oa = oriAverage;
			DescribePointSurfMod dm = describeMod;
			DescribePointSurf df = describeFast;
			os.setImage(ii); oa.setImage(ii);
#pragma omp for schedule(dynamic, 16)
			for (int i = 0; i < n; i++) {
				const ScalePoint& p = pts[i];
				double angle;
				if (stable) { os.setObjectRadius(p.scale); angle = os.compute(p.x, p.y); }
				else { oa.setObjectRadius(p.scale); angle = oa.compute(p.x, p.y); }
				double* d = &out.desc[(size_t)i * D * nb];
				DescribePointSurf& de = stable ? (DescribePointSurf&)dm : df;
				for (int b = 0; b < nb; b++) {
					de.setImage(bandII[b]);
					de.describeTuple(p.x, p.y, angle, p.scale, d + (size_t)b * D);
				}
				normalizeL2(d, D * nb);
				de.setImage(ii);
				out.white[i] = de.computeLaplaceSign((int)(p.x + 0.5), (int)(p.y + 0.5), p.scale) ? 1 : 0;
				out.angles[i] = angle;
			}
		}
	}
	void describeAll(const std::vector<ScalePoint>& pts, SurfResult& out) {
		const int n = (int)pts.size(), D = dof();
		out.points = pts;
		out.angles.assign(n, 0);
		out.desc.assign((size_t)n * D, 0);
		out.white.assign(n, 0);
#pragma omp parallel num_threads(threads) if (threads > 1)
		{
			// per-thread copy() of orientation + describe, as the _MT wrapper does
			OrientationSlidingWindow os = oriSliding;
			OrientationAverage oa = oriAverage;
			DescribePointSurfMod dm = describeMod;
			DescribePointSurf df = describeFast;
			os.setImage(ii); oa.setImage(ii); dm.setImage(ii); df.setImage(ii);
			os.setIntPixels(intPixels); oa.setIntPixels(intPixels); dm.setIntPixels(intPixels); df.setIntPixels(intPixels);
			BrightFeature bf;
#pragma omp for schedule(dynamic, 16)
			for (int i = 0; i < n; i++) {
				const ScalePoint& p = pts[i];
				double radius = p.scale * 2.0;  // BoofDefaults.SURF_SCALE_TO_RADIUS
				double angle;
				if (stable) { os.setObjectRadius(radius); angle = os.compute(p.x, p.y); }
				else { oa.setObjectRadius(radius); angle = oa.compute(p.x, p.y); }
				if (stable) dm.describe(p.x, p.y, angle, p.scale, bf);
				else df.describe(p.x, p.y, angle, p.scale, bf);
				out.angles[i] = angle;
				std::memcpy(&out.desc[(size_t)i * D], bf.value.data(), sizeof(double) * D);
				out.white[i] = bf.white ? 1 : 0;
			}
		}
	}
};

// ------------------------------------------------------------------------------------------------
// Association   F:alg/feature/associate/AssociateGreedy.java ; F:alg/descriptor/DescriptorDistance.java
// ------------------------------------------------------------------------------------------------
// DescriptorDistance.java:55-64
inline double euclideanSq(const double* a, const double* b, int N) {
	double total = 0;
	for (int i = 0; i < N; i++) {
		double d = a[i] - b[i];
		total += d * d;
	}
	return total;
}
// DescriptorDistance.java:213-220 hamming(int) (bit-twiddling form, Java int semantics)
inline int hammingWord(int32_t val) {
	int32_t v = val;
	v = v - ((v >> 1) & 0x55555555);
	v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
	int32_t c = (int32_t)((uint32_t)((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101u) >> 24;
	return c;
}
// DescriptorDistance.java:196-203
inline int hamming(const int32_t* a, const int32_t* b, int N) {
	int score = 0;
	for (int i = 0; i < N; i++) score += hammingWord(a[i] ^ b[i]);
	return score;
}

// AssociateGreedy.java:65-118 ; score(i,j) supplies the fit.  threads>1: AssociateGreedy_MT (src rows in parallel).
template <class Score>
inline void associateGreedy(int ns, int nd, Score score, double maxFitError, bool backwardsValidation, int* pairs, double* fitQuality,
							std::vector<double>& workBuffer, int threads = 1) {
	workBuffer.resize((size_t)ns * nd);
	(void)threads;
#pragma omp parallel for num_threads(threads) if (threads > 1) schedule(static)
	for (int i = 0; i < ns; i++) {
		double bestScore = maxFitError;
		int bestIndex = -1;
		size_t workIdx = (size_t)i * nd;
		for (int j = 0; j < nd; j++) {
			double fit = score(i, j);
			workBuffer[workIdx + j] = fit;
			if (fit <= bestScore) { bestIndex = j; bestScore = fit; }
		}
		pairs[i] = bestIndex;
		fitQuality[i] = bestScore;
	}
	if (backwardsValidation) {
#pragma omp parallel for num_threads(threads) if (threads > 1) schedule(static)
		for (int i = 0; i < ns; i++) {
			int match = pairs[i];
			if (match == -1) continue;
			double scoreToBeat = workBuffer[(size_t)i * nd + match];
			size_t m = match;
			for (int j = 0; j < ns; j++, m += nd) {
				if (workBuffer[m] <= scoreToBeat && j != i) {
					pairs[i] = -1;
					fitQuality[i] = DBL_MAX;
					break;
				}
			}
		}
	}
}

}  // namespace oracle
