// TEST INFRASTRUCTURE ONLY -- CPU restatement of the boofcv-ip front-end ops (convolution, Gaussian blur,
// Sobel / three-tap gradient, sub-sampling) and of the BRIEF descriptor.  See boof_oracle.hpp for the rules.
#pragma once
#include "boof_oracle.hpp"

namespace oracle {

// ------------------------------------------------------------------------------------------------
// 1D convolution without border   I:alg/filter/convolve/ConvolveImageNoBorder.java:53-77
// Unrolled (widths 3,5,7,9,11, symmetric offset): first tap assigns, the rest accumulate
//   I:alg/filter/convolve/noborder/ConvolveImageUnrolled_SB_F32_F32.java:50-150,152-180,347-382
// Standard: total = 0 then accumulate   I:alg/filter/convolve/noborder/ConvolveImageStandard_SB.java:44-104
// ------------------------------------------------------------------------------------------------
inline bool convIsUnrolled(int kw, int koff) {
	if (koff != kw / 2 || kw % 2 == 0) return false;
	return kw == 3 || kw == 5 || kw == 7 || kw == 9 || kw == 11;
}
inline void convolveHorizontalNoBorder(const float* ker, int kw, int koff, const GrayF32& image, GrayF32& dest, int threads = 1) {
	const bool unrolled = convIsUnrolled(kw, koff);
	const int width = image.width;
	(void)threads;
#pragma omp parallel for num_threads(threads) if (threads > 1) schedule(static)
	for (int i = 0; i < image.height; i++) {
		int indexDst = dest.startIndex + i * dest.stride + koff;
		int j = image.startIndex + i * image.stride;
		const int jEnd = j + width - (kw - 1);
		for (; j < jEnd; j++) {
			float total;
			if (unrolled) {
				total = image.data[j] * ker[0];
				for (int k = 1; k < kw; k++) total += image.data[j + k] * ker[k];
			} else {
				total = 0;
				for (int k = 0; k < kw; k++) total += image.data[j + k] * ker[k];
			}
			dest.data[indexDst++] = total;
		}
	}
}
inline void convolveVerticalNoBorder(const float* ker, int kw, int koff, const GrayF32& image, GrayF32& dest, int threads = 1) {
	const bool unrolled = convIsUnrolled(kw, koff);
	const int imgWidth = dest.width, imgHeight = dest.height;
	const int yEnd = imgHeight - (kw - koff - 1);
	(void)threads;
#pragma omp parallel for num_threads(threads) if (threads > 1) schedule(static)
	for (int y = koff; y < yEnd; y++) {
		int indexDst = dest.startIndex + y * dest.stride;
		int i = image.startIndex + (y - koff) * image.stride;
		const int iEnd = i + imgWidth;
		for (; i < iEnd; i++) {
			float total;
			int indexSrc = i;
			if (unrolled) {
				total = image.data[indexSrc] * ker[0];
				for (int k = 1; k < kw; k++) { indexSrc += image.stride; total += image.data[indexSrc] * ker[k]; }
			} else {
				total = 0;
				for (int k = 0; k < kw; k++) { total += image.data[indexSrc] * ker[k]; indexSrc += image.stride; }
			}
			dest.data[indexDst++] = total;
		}
	}
}

// ------------------------------------------------------------------------------------------------
// Normalised-border convolution   I:alg/filter/convolve/ConvolveImageNormalized.java:48-93
// ------------------------------------------------------------------------------------------------
// I:alg/filter/convolve/normalized/ConvolveNormalized_JustBorder_SB.java:42-90
inline void convNormBorderHorizontal(const float* ker, int kw, int offsetL, const GrayF32& input, GrayF32& output) {
	const int offsetR = kw - offsetL - 1;
	const int width = input.width, height = input.height;
	for (int i = 0; i < height; i++) {
		int indexDest = output.startIndex + i * output.stride;
		int j = input.startIndex + i * input.stride;
		const int jStart = j;
		int jEnd = j + offsetL;
		for (; j < jEnd; j++) {
			float total = 0, weight = 0;
			int indexSrc = jStart;
			for (int k = kw - (offsetR + 1 + j - jStart); k < kw; k++) {
				float w = ker[k];
				weight += w;
				total += input.data[indexSrc++] * w;
			}
			output.data[indexDest++] = total / weight;
		}
		j += width - (offsetL + offsetR);
		indexDest += width - (offsetL + offsetR);
		jEnd = jStart + width;
		for (; j < jEnd; j++) {
			float total = 0, weight = 0;
			int indexSrc = j - offsetL;
			const int kEnd = jEnd - indexSrc;
			for (int k = 0; k < kEnd; k++) {
				float w = ker[k];
				weight += w;
				total += input.data[indexSrc++] * w;
			}
			output.data[indexDest++] = total / weight;
		}
	}
}
// :92-145
inline void convNormBorderVertical(const float* ker, int kw, int offsetL, const GrayF32& input, GrayF32& output) {
	const int offsetR = kw - offsetL - 1;
	const int imgWidth = output.width, imgHeight = output.height;
	const int yEnd = imgHeight - offsetR;
	for (int y = 0; y < offsetL; y++) {
		int indexDst = output.startIndex + y * output.stride;
		int i = input.startIndex + y * input.stride;
		const int iEnd = i + imgWidth;
		int kStart = offsetL - y;
		float weight = 0;
		for (int k = kStart; k < kw; k++) weight += ker[k];
		for (; i < iEnd; i++) {
			float total = 0;
			int indexSrc = i - y * input.stride;
			for (int k = kStart; k < kw; k++, indexSrc += input.stride) total += input.data[indexSrc] * ker[k];
			output.data[indexDst++] = total / weight;
		}
	}
	for (int y = yEnd; y < imgHeight; y++) {
		int indexDst = output.startIndex + y * output.stride;
		int i = input.startIndex + y * input.stride;
		const int iEnd = i + imgWidth;
		int kEnd = imgHeight - (y - offsetL);
		float weight = 0;
		for (int k = 0; k < kEnd; k++) weight += ker[k];
		for (; i < iEnd; i++) {
			float total = 0;
			int indexSrc = i - offsetL * input.stride;
			for (int k = 0; k < kEnd; k++, indexSrc += input.stride) total += input.data[indexSrc] * ker[k];
			output.data[indexDst++] = total / weight;
		}
	}
}
// I:alg/filter/convolve/normalized/ConvolveNormalizedNaive_SB.java:37-84 (used when the kernel is wider than the image)
inline void convNormNaiveHorizontal(const float* ker, int kw, int offset, const GrayF32& input, GrayF32& output) {
	for (int y = 0; y < input.height; y++)
		for (int x = 0; x < input.width; x++) {
			float total = 0, weight = 0;
			int startX = x - offset, endX = startX + kw;
			if (startX < 0) startX = 0;
			if (endX > input.width) endX = input.width;
			for (int j = startX; j < endX; j++) {
				float v = ker[j - x + offset];
				total += input.get(j, y) * v;
				weight += v;
			}
			output.set(x, y, total / weight);
		}
}
inline void convNormNaiveVertical(const float* ker, int kw, int offset, const GrayF32& input, GrayF32& output) {
	for (int y = 0; y < input.height; y++)
		for (int x = 0; x < input.width; x++) {
			float total = 0, weight = 0;
			int startY = y - offset, endY = startY + kw;
			if (startY < 0) startY = 0;
			if (endY > input.height) endY = input.height;
			for (int i = startY; i < endY; i++) {
				float v = ker[i - y + offset];
				total += input.get(x, i) * v;
				weight += v;
			}
			output.set(x, y, total / weight);
		}
}
// ConvolveImageNormalized.java:48-68: re-normalise the kernel when |sum-1| > 1e-4 (Kernel1D_F32.computeSum: sequential)
inline std::vector<float> convNormKernel(const float* ker, int kw) {
	std::vector<float> k(ker, ker + kw);
	float sum = 0;
	for (int i = 0; i < kw; i++) sum += k[i];
	if (std::abs(sum - 1.0f) > 1e-4f) {
		float total = 0;
		for (int i = 0; i < kw; i++) total += k[i];
		for (int i = 0; i < kw; i++) k[i] /= total;
	}
	return k;
}
inline void convolveNormalizedHorizontal(const float* ker, int kw, int koff, const GrayF32& src, GrayF32& dst, int threads = 1) {
	if (kw >= src.width) { convNormNaiveHorizontal(ker, kw, koff, src, dst); return; }
	std::vector<float> k = convNormKernel(ker, kw);
	convolveHorizontalNoBorder(k.data(), kw, koff, src, dst, threads);
	convNormBorderHorizontal(k.data(), kw, koff, src, dst);
}
inline void convolveNormalizedVertical(const float* ker, int kw, int koff, const GrayF32& src, GrayF32& dst, int threads = 1) {
	if (kw >= src.height) { convNormNaiveVertical(ker, kw, koff, src, dst); return; }
	std::vector<float> k = convNormKernel(ker, kw);
	convolveVerticalNoBorder(k.data(), kw, koff, src, dst, threads);
	convNormBorderVertical(k.data(), kw, koff, src, dst);
}
// I:alg/filter/blur/BlurImageOps.java:406-425 gaussian(GrayF32,out,sigma,radius,storage) (sigmaX==sigmaY form)
inline void blurGaussian(const GrayF32& input, GrayF32& output, double sigma, int radius, GrayF32& storage, int threads = 1) {
	Kernel1D_F32 k = gaussian1D_F32(sigma, radius);
	convolveNormalizedHorizontal(k.data.data(), k.width, k.offset, input, storage, threads);
	convolveNormalizedVertical(k.data.data(), k.width, k.offset, storage, output, threads);
}

// ------------------------------------------------------------------------------------------------
// Gradients.  borderZero=false: border pixels untouched (border == null); true: ImageBorderValue(0)
// ------------------------------------------------------------------------------------------------
// I:alg/filter/derivative/impl/GradientSobel_UnrolledOuter.java:210- (register rotation unrolled away; same expressions)
inline void gradientSobel(const GrayF32& orig, GrayF32& derivX, GrayF32& derivY, bool borderZero, int threads = 1) {
	const float* data = orig.data;
	const int width = orig.width, height = orig.height, s = orig.stride;
	(void)threads;
#pragma omp parallel for num_threads(threads) if (threads > 1) schedule(static)
	for (int y = 1; y < height - 1; y++) {
		int index = orig.startIndex + s * y + 1;
		int indexX = derivX.startIndex + derivX.stride * y + 1;
		int indexY = derivY.startIndex + derivY.stride * y + 1;
		for (int x = 1; x < width - 1; x++, index++) {
			float a11 = data[index - s - 1], a12 = data[index - s], a13 = data[index - s + 1];
			float a21 = data[index - 1], a23 = data[index + 1];
			float a31 = data[index + s - 1], a32 = data[index + s], a33 = data[index + s + 1];
			float v = (a33 - a11) * 0.25f;
			float w = (a31 - a13) * 0.25f;
			derivY.data[indexY++] = (a32 - a12) * 0.5f + v + w;
			derivX.data[indexX++] = (a23 - a21) * 0.5f + v - w;
		}
	}
	if (borderZero) {
		// I:alg/filter/convolve/border/ConvolveJustBorder_General_SB.java:110-175 with GradientSobel.kernelDerivX/Y_F32 (:75-78)
		static const float kx[9] = {-0.25f, 0, 0.25f, -0.5f, 0, 0.5f, -0.25f, 0, 0.25f};
		static const float ky[9] = {-0.25f, -0.5f, -0.25f, 0, 0, 0, 0.25f, 0.5f, 0.25f};
		auto at = [&](int x, int y) -> float { return orig.isInBounds(x, y) ? orig.get(x, y) : 0.0f; };
		auto px = [&](int x, int y) {
			float tx = 0, ty = 0;
			int ik = 0;
			for (int i = -1; i <= 1; i++)
				for (int j = -1; j <= 1; j++, ik++) { tx += at(x + j, y + i) * kx[ik]; ty += at(x + j, y + i) * ky[ik]; }
			derivX.set(x, y, tx);
			derivY.set(x, y, ty);
		};
		for (int y = 0; y < height; y++) { px(0, y); if (width > 1) px(width - 1, y); }
		for (int x = 1; x < width - 1; x++) { px(x, 0); if (height > 1) px(x, height - 1); }
	}
}
// I:alg/filter/derivative/impl/GradientThree_Standard.java:40-62
inline void gradientThree(const GrayF32& orig, GrayF32& derivX, GrayF32& derivY, bool borderZero, int threads = 1) {
	const float* data = orig.data;
	const int width = orig.width, height = orig.height, s = orig.stride;
	(void)threads;
#pragma omp parallel for num_threads(threads) if (threads > 1) schedule(static)
	for (int y = 1; y < height - 1; y++) {
		int indexX = derivX.startIndex + derivX.stride * y + 1;
		int indexY = derivY.startIndex + derivY.stride * y + 1;
		int indexSrc = orig.startIndex + s * y + 1;
		const int endX = indexSrc + width - 2;
		for (; indexSrc < endX; indexSrc++) {
			derivX.data[indexX++] = (data[indexSrc + 1] - data[indexSrc - 1]) * 0.5f;
			derivY.data[indexY++] = (data[indexSrc + s] - data[indexSrc - s]) * 0.5f;
		}
	}
	if (borderZero) {
		// I:alg/filter/derivative/DerivativeHelperFunctions.java:140-175: border columns/rows through the generic
		// border convolution (total=0; total += get*k ...), the remaining first/last rows (for X) and columns (for Y)
		// through the unrolled 3-tap no-border convolution with kernel {-0.5,0,0.5}
		auto at = [&](int x, int y) -> float { return orig.isInBounds(x, y) ? orig.get(x, y) : 0.0f; };
		static const float k[3] = {-0.5f, 0, 0.5f};
		auto genX = [&](int x, int y) { float t = 0; for (int i = 0; i < 3; i++) t += at(x + i - 1, y) * k[i]; return t; };
		auto genY = [&](int x, int y) { float t = 0; for (int i = 0; i < 3; i++) t += at(x, y + i - 1) * k[i]; return t; };
		auto unrX = [&](int x, int y) { float t = orig.get(x - 1, y) * k[0]; t += orig.get(x, y) * k[1]; t += orig.get(x + 1, y) * k[2]; return t; };
		auto unrY = [&](int x, int y) { float t = orig.get(x, y - 1) * k[0]; t += orig.get(x, y) * k[1]; t += orig.get(x, y + 1) * k[2]; return t; };
		for (int y = 0; y < height; y++) { derivX.set(0, y, genX(0, y)); derivX.set(width - 1, y, genX(width - 1, y)); }
		for (int x = 0; x < width; x++) { derivY.set(x, 0, genY(x, 0)); derivY.set(x, height - 1, genY(x, height - 1)); }
		// sub-images of the first two and last two rows (X) / columns (Y); their interior overlaps rows 1 and h-2,
		// which the main loop already wrote with the equivalent (b-a)*0.5 expression -- the reference overwrites them
		for (int x = 1; x < width - 1; x++) {
			for (int y : {0, 1, height - 2, height - 1}) if (y >= 0 && y < height) derivX.set(x, y, unrX(x, y));
		}
		for (int y = 1; y < height - 1; y++) {
			for (int x : {0, 1, width - 2, width - 1}) if (x >= 0 && x < width) derivY.set(x, y, unrY(x, y));
		}
	}
}
// integer sub-sampling of a layer (pyramid down-sample step): out(x,y) = in(x*skip, y*skip)
inline void pyramidSubsample(const GrayF32& in, GrayF32& out, int skip) {
	for (int y = 0; y < out.height; y++)
		for (int x = 0; x < out.width; x++) out.set(x, y, in.get(x * skip, y * skip));
}

// ------------------------------------------------------------------------------------------------
// BRIEF   F:alg/feature/describe/brief/FactoryBriefDefinition.java, DescribePointBinaryCompare.java,
//         impl/ImplDescribeBinaryCompare_F32.java
// ------------------------------------------------------------------------------------------------
struct Point2D_I32 { int x, y; };
struct BinaryCompareDefinition {
	int radius = 0;
	std::vector<Point2D_I32> samplePoints;
	std::vector<Point2D_I32> compare;
};
// FactoryBriefDefinition.java:73-85 randomGaussian
inline Point2D_I32 briefRandomGaussian(JavaRandom& rand, double sigma, int radius) {
	int x, y;
	while (true) {
		x = (int)(rand.nextGaussian() * sigma);
		y = (int)(rand.nextGaussian() * sigma);
		if (std::sqrt((double)(x * x + y * y)) < radius) break;
	}
	return {x, y};
}
// FactoryBriefDefinition.java:57-67 gaussian2
inline BinaryCompareDefinition briefGaussian2(JavaRandom& rand, int radius, int numPairs) {
	BinaryCompareDefinition ret;
	ret.radius = radius;
	ret.samplePoints.resize(numPairs);
	ret.compare.resize(numPairs);
	double sigma = (2.0 * radius + 1.0) / 5.0;
	for (int i = 0; i < numPairs; i++) {
		ret.samplePoints[i] = briefRandomGaussian(rand, sigma, radius);
		ret.compare[i] = {i, rand.nextInt(numPairs)};
	}
	return ret;
}
struct DescribeBinaryCompare_F32 {
	BinaryCompareDefinition def;
	const GrayF32* image = nullptr;
	std::vector<int> offsetsA, offsetsB;
	explicit DescribeBinaryCompare_F32(const BinaryCompareDefinition& d) : def(d) {}
	// DescribePointBinaryCompare.java:67-82
	void setImage(const GrayF32& img) {
		image = &img;
		std::vector<int> offsets(def.samplePoints.size());
		for (size_t i = 0; i < offsets.size(); i++) offsets[i] = img.stride * def.samplePoints[i].y + def.samplePoints[i].x;
		offsetsA.resize(def.compare.size());
		offsetsB.resize(def.compare.size());
		for (size_t i = 0; i < def.compare.size(); i++) { offsetsA[i] = offsets[def.compare[i].x]; offsetsB[i] = offsets[def.compare[i].y]; }
	}
	// DescribePointBinaryCompare.java:90-98 ; BoofMiscOps.checkInside T:misc/BoofMiscOps.java:200-211
	void process(int c_x, int c_y, int32_t* feature) const {
		const GrayF32& im = *image;
		bool inside = !(c_x - def.radius < 0 || c_x + def.radius >= im.width || c_y - def.radius < 0 || c_y + def.radius >= im.height);
		const int n = (int)def.compare.size();
		const int words = (n + 31) / 32;
		for (int i = 0; i < words; i++) feature[i] = 0;
		int index = im.startIndex + im.stride * c_y + c_x;
		if (inside) {
			// ImplDescribeBinaryCompare_F32.java:47-71
			for (int i = 0; i < n; i += 32) {
				int end = std::min(n, i + 32);
				uint32_t desc = im.data[index + offsetsA[i]] < im.data[index + offsetsB[i]] ? 1 : 0;
				for (int j = i + 1; j < end; j++) {
					desc *= 2;
					if (im.data[index + offsetsA[j]] < im.data[index + offsetsB[j]]) desc += 1;
				}
				feature[i / 32] = (int32_t)desc;
			}
		} else {
			// :74-101
			for (int i = 0; i < n; i += 32) {
				int end = std::min(n, i + 32);
				uint32_t desc = 0;
				for (int j = i; j < end; j++) {
					Point2D_I32 c = def.compare[j];
					Point2D_I32 p_a = def.samplePoints[c.x], p_b = def.samplePoints[c.y];
					if (im.isInBounds(p_a.x + c_x, p_a.y + c_y) && im.isInBounds(p_b.x + c_x, p_b.y + c_y)) {
						desc *= 2;
						if (im.data[index + offsetsA[j]] < im.data[index + offsetsB[j]]) desc += 1;
					}
				}
				feature[i / 32] = (int32_t)desc;
			}
		}
	}
};

}  // namespace oracle
