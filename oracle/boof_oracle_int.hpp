// TEST INFRASTRUCTURE ONLY -- CPU restatement of the integer-image variants at stage level:
//   IntegralImageOps.transform(GrayU8, GrayS32), the Fast-Hessian intensity on a GrayS32 integral image, BRIEF on GrayU8.
// See boof_oracle.hpp for the rules.  Integer box sums are exact; the places where the reference converts to float are kept.
#pragma once
#include "boof_oracle.hpp"

namespace oracle {

struct GrayU8v { const uint8_t* data; int startIndex, stride, width, height; };
struct GrayS32v { int32_t* data; int startIndex, stride, width, height; };

// I:alg/transform/ii/impl/ImplIntegralImageOps.java:94-118  transform(GrayU8, GrayS32)
inline void integral_transform_u8(const GrayU8v& input, GrayS32v& transformed) {
	int indexSrc = input.startIndex, indexDst = transformed.startIndex;
	int end = indexSrc + input.width;
	int32_t total = 0;
	for (; indexSrc < end; indexSrc++) transformed.data[indexDst++] = total += input.data[indexSrc] & 0xFF;
	for (int y = 1; y < input.height; y++) {
		indexSrc = input.startIndex + input.stride * y;
		indexDst = transformed.startIndex + transformed.stride * y;
		int indexPrev = indexDst - transformed.stride;
		end = indexSrc + input.width;
		total = 0;
		for (; indexSrc < end; indexSrc++) {
			total += input.data[indexSrc] & 0xFF;
			transformed.data[indexDst++] = transformed.data[indexPrev++] + total;
		}
	}
}
// :239-258 block_zero(GrayS32)
inline int32_t block_zero_s32(const GrayS32v& ii, int x0, int y0, int x1, int y1) {
	x0 = std::min(x0, ii.width - 1); y0 = std::min(y0, ii.height - 1); x1 = std::min(x1, ii.width - 1); y1 = std::min(y1, ii.height - 1);
	int32_t br = 0, tr = 0, bl = 0, tl = 0;
	if (x1 >= 0 && y1 >= 0) br = ii.data[ii.startIndex + y1 * ii.stride + x1];
	if (y0 >= 0 && x1 >= 0) tr = ii.data[ii.startIndex + y0 * ii.stride + x1];
	if (x0 >= 0 && y1 >= 0) bl = ii.data[ii.startIndex + y1 * ii.stride + x0];
	if (x0 >= 0 && y0 >= 0) tl = ii.data[ii.startIndex + y0 * ii.stride + x0];
	return br - tr - bl + tl;
}
// convolveSparse(GrayS32, IntegralKernel, x, y) -> int (ImplIntegralImageOps.java, S32 form)
inline int32_t convolveSparse_s32(const GrayS32v& ii, const IntegralKernel& k, int x, int y) {
	int32_t ret = 0;
	for (int i = 0; i < k.n; i++) ret += block_zero_s32(ii, x + k.x0[i], y + k.y0[i], x + k.x1[i], y + k.y1[i]) * k.scales[i];
	return ret;
}
// F:alg/feature/detect/intensity/impl/ImplIntegralImageFeatureIntensity.java:288-301 computeHessian(GrayS32): int -> float at the assignment
inline void computeHessian_s32(const GrayS32v& ii, GrayF32& intensity, const IntegralKernel& kerXX, const IntegralKernel& kerYY, const IntegralKernel& kerXY,
							   float norm, int y, int yy, int x, int xx) {
	float Dxx = (float)convolveSparse_s32(ii, kerXX, xx, yy);
	float Dyy = (float)convolveSparse_s32(ii, kerYY, xx, yy);
	float Dxy = (float)convolveSparse_s32(ii, kerXY, xx, yy);
	Dxx *= norm; Dxy *= norm; Dyy *= norm;
	intensity.set(x, y, Dxx * Dyy - 0.81f * Dxy * Dxy);
}
// :245-286 hessianBorder(GrayS32) + :305-390 hessianInner(GrayS32)
inline void hessian_s32(const GrayS32v& ii, int skip, int size, GrayF32& intensity) {
	const int w = intensity.width, h = intensity.height;
	IntegralKernel kerXX = kernelDerivXX(size), kerYY = kernelDerivYY(size), kerXY = kernelDerivXY(size);
	const int radiusFeature = size / 2;
	const int borderOrig = radiusFeature + 1 + (skip - (radiusFeature + 1) % skip);
	const int border = borderOrig / skip;
	const float norm = 1.0f / (size * size);
	for (int y = 0; y < h; y++) {
		int yy = y * skip;
		for (int x = 0; x < border; x++) computeHessian_s32(ii, intensity, kerXX, kerYY, kerXY, norm, y, yy, x, x * skip);
		for (int x = w - border; x < w; x++) computeHessian_s32(ii, intensity, kerXX, kerYY, kerXY, norm, y, yy, x, x * skip);
	}
	for (int x = border; x < w - border; x++) {
		int xx = x * skip;
		for (int y = 0; y < border; y++) computeHessian_s32(ii, intensity, kerXX, kerYY, kerXY, norm, y, y * skip, x, xx);
		for (int y = h - border; y < h; y++) computeHessian_s32(ii, intensity, kerXX, kerYY, kerXY, norm, y, y * skip, x, xx);
	}
	// inner: int box sums, converted where the Java assigns / compounds into a float
	const int blockSmall = size / 3, blockLarge = size - blockSmall - 1, radiusSkinny = blockLarge / 2;
	const int blockW2 = 2 * blockSmall, blockW3 = 3 * blockSmall;
	const int rowOff1 = blockSmall * ii.stride, rowOff2 = 2 * rowOff1, rowOff3 = 3 * rowOff1;
	const int lostPixel = borderOrig - radiusFeature - 1;
	const int endY = h - border, endX = w - border;
	const int32_t* d = ii.data;
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
			float Dxx = (float)(d[indexBottom + blockW3] - d[indexTop + blockW3] - d[indexBottom] + d[indexTop]);
			Dxx -= (float)(3 * (d[indexBottom + blockW2] - d[indexTop + blockW2] - d[indexBottom + blockSmall] + d[indexTop + blockSmall]));
			float Dyy = (float)(d[indexR + rowOff3] - d[indexL + rowOff3] - d[indexR] + d[indexL]);
			Dyy -= (float)(3 * (d[indexR + rowOff2] - d[indexL + rowOff2] - d[indexR + rowOff1] + d[indexL + rowOff1]));
			int x3 = blockSmall + 1, x4 = x3 + blockSmall;
			float Dxy = (float)(d[indexY2 + blockSmall] - d[indexY1 + blockSmall] - d[indexY2] + d[indexY1]);
			Dxy -= (float)(d[indexY2 + x4] - d[indexY1 + x4] - d[indexY2 + x3] + d[indexY1 + x3]);
			Dxy += (float)(d[indexY4 + x4] - d[indexY3 + x4] - d[indexY4 + x3] + d[indexY3 + x3]);
			Dxy -= (float)(d[indexY4 + blockSmall] - d[indexY3 + blockSmall] - d[indexY4] + d[indexY3]);
			Dxx *= norm; Dxy *= norm; Dyy *= norm;
			intensity.data[indexDst] = Dxx * Dyy - 0.81f * Dxy * Dxy;
			indexTop += skip; indexBottom += skip; indexL += skip; indexR += skip;
			indexY1 += skip; indexY2 += skip; indexY3 += skip; indexY4 += skip;
		}
	}
}

// the overload FastHessianFeatureDetector::detectOctave<GrayS32v> resolves to
inline void hessian(const GrayS32v& ii, int skip, int size, GrayF32& intensity, int /*threads*/ = 1) { hessian_s32(ii, skip, size, intensity); }

// FactoryDetectDescribe.surfStable / surfFast on GrayU8 (integral type GrayS32): WrapDetectDescribeSurf.detect with
// GIntegralImageOps.transform(GrayU8, GrayS32), FastHessianFeatureDetector<GrayS32>, orientation and descriptor on I32 gradients
inline void surfDetectU8(DetectDescribeSurf& dd, const GrayU8v& input, SurfResult& out) {
	dd.ii.reshape(input.width, input.height);   // float storage reused as 32-bit words
	GrayS32v v{reinterpret_cast<int32_t*>(dd.ii.data), dd.ii.startIndex, dd.ii.stride, input.width, input.height};
	integral_transform_u8(input, v);
	dd.intPixels = true;
	dd.detector.threads = dd.threads;
	dd.detector.detect(v);
	dd.describeAll(dd.detector.foundPoints, out);
}

// F:alg/feature/describe/impl/ImplDescribeBinaryCompare_U8.java:47-101 (+ DescribePointBinaryCompare.process :67-98): unlike the F32
// class, the border form shifts the word for EVERY pair, in bounds or not
inline void brief_u8(const GrayU8v& image, int radius, int numPoints, const int* samplePoints /*[n][2]*/, const int* compare /*[n][2]*/, int c_x, int c_y,
					 int32_t* out /*ceil(n/32)*/) {
	const bool inside = !(c_x - radius < 0 || c_x + radius >= image.width || c_y - radius < 0 || c_y + radius >= image.height);
	const int index = image.startIndex + image.stride * c_y + c_x;
	for (int i = 0; i < numPoints; i += 32) {
		const int end = std::min(numPoints, i + 32);
		int32_t desc = 0;
		for (int j = i; j < end; j++) {
			const int ax = samplePoints[2 * compare[2 * j]], ay = samplePoints[2 * compare[2 * j] + 1];
			const int bx = samplePoints[2 * compare[2 * j + 1]], by = samplePoints[2 * compare[2 * j + 1] + 1];
			const int offA = ay * image.stride + ax, offB = by * image.stride + bx;
			if (inside) {
				int valA = image.data[index + offA] & 0xFF, valB = image.data[index + offB] & 0xFF;
				if (j == i) desc = valA < valB ? 1 : 0;
				else { desc = (int32_t)((uint32_t)desc * 2u); if (valA < valB) desc += 1; }
			} else {
				desc = (int32_t)((uint32_t)desc * 2u);
				const bool inA = ax + c_x >= 0 && ax + c_x < image.width && ay + c_y >= 0 && ay + c_y < image.height;
				const bool inB = bx + c_x >= 0 && bx + c_x < image.width && by + c_y >= 0 && by + c_y < image.height;
				if (inA && inB) {
					int valA = image.data[index + offA] & 0xFF, valB = image.data[index + offB] & 0xFF;
					if (valA < valB) desc += 1;
				}
			}
		}
		out[i / 32] = desc;
	}
}

}  // namespace oracle
