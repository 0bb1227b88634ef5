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
// Down-sampling convolution with a re-normalised border  (pyramid layer step)
//   I:alg/filter/convolve/ConvolveImageDownNormalized.java:53-86  (dispatch: naive when kernel.width >= image.width,
//     for BOTH directions -- the vertical form also tests image.width; else no-border interior, then just-border)
//   I:alg/filter/convolve/down/UtilDownConvolve.java:27-44
// ------------------------------------------------------------------------------------------------
inline int downComputeMaxSide(int sideLength, int skip, int radius) {
	int ret = sideLength - (sideLength % skip);
	if (ret + radius >= sideLength) {
		ret = sideLength - radius - 1;
		ret = ret - (ret % skip);
	} else {
		ret -= skip;
	}
	return ret;
}
inline int downComputeOffset(int skip, int radius) { return radius <= skip ? skip : radius + radius % skip; }

struct DownConvError : std::runtime_error { using std::runtime_error::runtime_error; };
// bounds-checked read standing in for Java's array check (Java throws ArrayIndexOutOfBounds when a loop
// leaves the backing array; leaving only the ROW would read a neighbouring row -- both are flagged here,
// the product returns BHIP_ERR_INVALID for the same shapes)
inline float downRead(const GrayF32& im, int x, int y) {
	if (x < 0 || y < 0 || x >= im.width || y >= im.height) throw DownConvError("down-convolution reads outside the image");
	return im.data[im.startIndex + y * im.stride + x];
}
// I:alg/filter/convolve/ConvolveImageDownNoBorder.java:160-176
inline void downCheckH(const GrayF32& in, const GrayF32& out, int skip) {
	if (skip <= 0) throw DownConvError("Skip must be >= 1");
	if (out.width < in.width / skip) throw DownConvError("Output width is too small");
	if (out.height < in.height) throw DownConvError("Output height is too small");
}
inline void downCheckV(const GrayF32& in, const GrayF32& out, int skip) {
	if (skip <= 0) throw DownConvError("Skip must be >= 1");
	if (out.width < in.width) throw DownConvError("Output width is too small");
	if (out.height < in.height / skip) throw DownConvError("Output height is too small");
}
// ConvolveImageDownNormalized.checkParameters == ConvolveImageDownNoBorder.checkParameters (:178-186)
inline void downCheck(const GrayF32& in, const GrayF32& out, int skip) {
	if (skip <= 0) throw DownConvError("Skip must be >= 1");
	if (out.width < in.width / skip) throw DownConvError("Output width is too small");
	if (out.height < in.height / skip) throw DownConvError("Output height is too small");
}
// interior: unrolled widths 3..11 assign the first tap (ConvolveDownNoBorderUnrolled_F32_F32.java:140-154, 351-366),
// every other width starts from 0 (ConvolveDownNoBorderStandard.java:43-77, 79-110)
inline void downNoBorderHorizontal(const float* ker, int kw, const GrayF32& input, GrayF32& output, int skip) {
	if (kw % 2 != 1) throw DownConvError("Non symmetric odd kernels not supported");
	const int radius = kw / 2;
	const bool unrolled = convIsUnrolled(kw, radius);
	const int widthEnd = downComputeMaxSide(input.width, skip, radius);
	const int offsetX = downComputeOffset(skip, radius);
	for (int i = 0; i < input.height; i++) {
		int dstX = offsetX / skip;
		for (int x = offsetX; x <= widthEnd; x += skip) {
			float total;
			if (unrolled) {
				total = downRead(input, x - radius, i) * ker[0];
				for (int k = 1; k < kw; k++) total += downRead(input, x - radius + k, i) * ker[k];
			} else {
				total = 0;
				for (int k = 0; k < kw; k++) total += downRead(input, x - radius + k, i) * ker[k];
			}
			output.set(dstX++, i, total);
		}
	}
}
inline void downNoBorderVertical(const float* ker, int kw, const GrayF32& input, GrayF32& output, int skip) {
	if (kw % 2 != 1) throw DownConvError("Non symmetric odd kernels not supported");
	const int radius = kw / 2;
	const bool unrolled = convIsUnrolled(kw, radius);
	const int heightEnd = downComputeMaxSide(input.height, skip, radius);
	const int offsetY = downComputeOffset(skip, radius);
	for (int y = offsetY; y <= heightEnd; y += skip) {
		for (int x = 0; x < input.width; x++) {
			float total;
			if (unrolled) {
				total = downRead(input, x, y - radius) * ker[0];
				for (int k = 1; k < kw; k++) total += downRead(input, x, y - radius + k) * ker[k];
			} else {
				total = 0;
				for (int k = 0; k < kw; k++) total += downRead(input, x, y - radius + k) * ker[k];
			}
			output.set(x, y / skip, total);
		}
	}
}
// I:alg/filter/convolve/down/ConvolveDownNormalized_JustBorder.java:43-90
inline void downJustBorderHorizontal(const float* ker, int kw, const GrayF32& input, GrayF32& output, int skip) {
	const int radius = kw / 2;
	const int offset = downComputeOffset(skip, radius);
	const int offsetEnd = downComputeMaxSide(input.width, skip, radius) + skip;
	const int width = input.width - input.width % skip;
	for (int y = 0; y < input.height; y++) {
		int dstX = 0;
		for (int x = 0; x < offset; x += skip) {
			float total = 0, weight = 0;
			for (int k = -x; k <= radius; k++) {
				float w = ker[k + radius];
				weight += w;
				total += downRead(input, x + k, y) * w;
			}
			output.set(dstX++, y, total / weight);
		}
		dstX = offsetEnd / skip;
		for (int x = offsetEnd; x < width; x += skip) {
			float total = 0, weight = 0;
			int endKernel = input.width - x - 1;
			if (endKernel > radius) endKernel = radius;
			for (int k = -radius; k <= endKernel; k++) {
				float w = ker[k + radius];
				weight += w;
				total += downRead(input, x + k, y) * w;
			}
			output.set(dstX++, y, total / weight);
		}
	}
}
// :92-139
inline void downJustBorderVertical(const float* ker, int kw, const GrayF32& input, GrayF32& output, int skip) {
	const int radius = kw / 2;
	const int offset = downComputeOffset(skip, radius);
	const int offsetEnd = downComputeMaxSide(input.height, skip, radius) + skip;
	const int width = input.width;
	const int height = input.height - input.height % skip;
	for (int y = 0; y < offset; y += skip) {
		for (int x = 0; x < width; x++) {
			float total = 0, weight = 0;
			for (int k = -y; k <= radius; k++) {
				float w = ker[k + radius];
				weight += w;
				total += downRead(input, x, y + k) * w;
			}
			output.set(x, y / skip, total / weight);
		}
	}
	for (int y = offsetEnd; y < height; y += skip) {
		int endKernel = input.height - y - 1;
		if (endKernel > radius) endKernel = radius;
		for (int x = 0; x < width; x++) {
			float total = 0, weight = 0;
			for (int k = -radius; k <= endKernel; k++) {
				float w = ker[k + radius];
				weight += w;
				total += downRead(input, x, y + k) * w;
			}
			output.set(x, y / skip, total / weight);
		}
	}
}
// I:alg/filter/convolve/down/ConvolveDownNormalizedNaive.java:40-94
inline void downNaiveHorizontal(const float* ker, int kw, const GrayF32& input, GrayF32& output, int skip) {
	const int radius = kw / 2;
	const int width = input.width - input.width % skip;
	for (int y = 0; y < input.height; y++)
		for (int x = 0; x < width; x += skip) {
			float total = 0, div = 0;
			int startX = x - radius, endX = x + radius;
			if (startX < 0) startX = 0;
			if (endX >= input.width) endX = input.width - 1;
			for (int j = startX; j <= endX; j++) {
				float v = ker[j - x + radius];
				total += input.get(j, y) * v;
				div += v;
			}
			output.set(x / skip, y, total / div);
		}
}
inline void downNaiveVertical(const float* ker, int kw, const GrayF32& input, GrayF32& output, int skip) {
	const int radius = kw / 2;
	const int height = input.height - input.height % skip;
	for (int y = 0; y < height; y += skip)
		for (int x = 0; x < input.width; x++) {
			float total = 0, div = 0;
			int startY = y - radius, endY = y + radius;
			if (startY < 0) startY = 0;
			if (endY >= input.height) endY = input.height - 1;
			for (int i = startY; i <= endY; i++) {
				float v = ker[i - y + radius];
				total += input.get(x, i) * v;
				div += v;
			}
			output.set(x, y / skip, total / div);
		}
}
inline void convolveDownNormalizedHorizontal(const float* ker, int kw, const GrayF32& image, GrayF32& dest, int skip) {
	downCheck(image, dest, skip);
	if (kw >= image.width) {
		downCheckH(image, dest, skip);  // GrayF32.set(x,y,..) bounds check (ImageAccessException)
		downNaiveHorizontal(ker, kw, image, dest, skip);
	} else {
		downCheckH(image, dest, skip);
		downNoBorderHorizontal(ker, kw, image, dest, skip);
		downJustBorderHorizontal(ker, kw, image, dest, skip);
	}
}
inline void convolveDownNormalizedVertical(const float* ker, int kw, const GrayF32& image, GrayF32& dest, int skip) {
	downCheck(image, dest, skip);
	if (kw >= image.width) {  // sic: the reference tests the WIDTH here too
		downCheckV(image, dest, skip);  // GrayF32.set(x,y,..) bounds check (ImageAccessException)
		downNaiveVertical(ker, kw, image, dest, skip);
	} else {
		downCheckV(image, dest, skip);
		downNoBorderVertical(ker, kw, image, dest, skip);
		downJustBorderVertical(ker, kw, image, dest, skip);
	}
}

// ------------------------------------------------------------------------------------------------
// PyramidDiscreteSampleBlur   I:alg/transform/pyramid/PyramidDiscreteSampleBlur.java:68-126
//   layer sizes  T:struct/pyramid/ImagePyramidBase.java:73-95  (ceil(bottom/scale); freshly created images are zero,
//   the convolution writes floor(prev/skip) columns/rows, so a ceil-only last column/row stays 0)
//   factory  I:factory/transform/pyramid/FactoryPyramid.java:53-61
// ------------------------------------------------------------------------------------------------
struct PyramidDiscreteSampleBlur {
	std::vector<float> kernel;
	std::vector<int> scale;
	std::vector<double> sigmas;
	std::vector<GrayF32> layers;
	GrayF32 temp;
	int bottomWidth = 0, bottomHeight = 0;

	PyramidDiscreteSampleBlur(const float* ker, int kw, double sigma, const int* scaleFactors, int n) : kernel(ker, ker + kw), scale(scaleFactors, scaleFactors + n) {
		// PyramidDiscrete.setScaleFactors -> checkScales (ImagePyramidBase.java:100-112)
		if (n > 0 && scale[0] < 0) throw DownConvError("The first layer must be more than zero.");
		int prevScale = 0;
		for (int s : scale) { if (s < prevScale) throw DownConvError("Higher layers must be the same size or larger than previous layers."); prevScale = s; }
		sigmas.assign(n, 0.0);
		for (int i = 1; i < n; i++) {
			double prev = sigmas[i - 1];
			double applied = sigma * scale[i - 1];
			sigmas[i] = std::sqrt(prev * prev + applied * applied);
		}
	}
	void initialize(int width, int height) {
		if (bottomWidth == width && bottomHeight == height) return;
		bottomWidth = width; bottomHeight = height;
		layers.clear();
		layers.resize(scale.size());
		for (size_t i = 0; i < scale.size(); i++) {
			double scaleFactor = scale[i];
			int w = (int)std::ceil(bottomWidth / scaleFactor), h = (int)std::ceil(bottomHeight / scaleFactor);
			if (i == 0 && scale[0] == 1) { w = bottomWidth; h = bottomHeight; }
			layers[i].reshape(w, h);
			std::fill(layers[i].storage.begin(), layers[i].storage.end(), 0.f);
		}
	}
	void process(const GrayF32& input) {
		initialize(input.width, input.height);
		if (scale[0] == 1) {
			// setFirstLayer(input) / getLayer(0).setTo(input): identical pixel values either way
			for (int y = 0; y < input.height; y++)
				for (int x = 0; x < input.width; x++) layers[0].set(x, y, input.get(x, y));
		} else {
			int skip = scale[0];
			temp.reshape(input.width / skip, input.height);
			convolveDownNormalizedHorizontal(kernel.data(), (int)kernel.size(), input, temp, skip);
			convolveDownNormalizedVertical(kernel.data(), (int)kernel.size(), temp, layers[0], skip);
		}
		for (size_t index = 1; index < scale.size(); index++) {
			int skip = scale[index] / scale[index - 1];
			GrayF32& prev = layers[index - 1];
			temp.reshape(prev.width / skip, prev.height);
			convolveDownNormalizedHorizontal(kernel.data(), (int)kernel.size(), prev, temp, skip);
			convolveDownNormalizedVertical(kernel.data(), (int)kernel.size(), temp, layers[index], skip);
		}
	}
};

// ------------------------------------------------------------------------------------------------
// Remaining BOverride hooks of the boofcv-ip front end: 2-D convolution, mean blur, median blur
// ------------------------------------------------------------------------------------------------
// ConvolveImageNoBorder.convolve(Kernel2D_F32, GrayF32, GrayF32)  I:alg/filter/convolve/ConvolveImageNoBorder.java:79-90
//   unrolled (odd symmetric widths 3..11): every kernel row is summed from 0 on its own and the row sums are added in order
//     I:alg/filter/convolve/noborder/ConvolveImageUnrolled_SB_F32_F32.java:118-150, 592-644
//   standard: one running total over the kernel in row-major order   I:.../noborder/ConvolveImageStandard_SB.java:106-134
// The frame of `dest` (offset pixels) is left untouched.
inline void convolve2DNoBorder(const float* ker, int kw, int koff, const GrayF32& src, GrayF32& dest) {
	const bool unrolled = convIsUnrolled(kw, koff);
	const int width = src.width, height = src.height;
	const int offsetL = koff, offsetR = kw - koff - 1;
	for (int y = offsetL; y < height - offsetR; y++)
		for (int x = offsetL; x < width - offsetR; x++) {
			float total = 0;
			if (unrolled) {
				for (int i = 0; i < kw; i++) {
					float rowTotal = 0;
					for (int j = 0; j < kw; j++) rowTotal += src.get(x - offsetL + j, y - offsetL + i) * ker[i * kw + j];
					if (i == 0) total = rowTotal; else total += rowTotal;
				}
			} else {
				int indexKer = 0;
				for (int ki = 0; ki < kw; ki++)
					for (int kj = 0; kj < kw; kj++) total += src.get(x - offsetL + kj, y + ki - offsetL) * ker[indexKer++];
			}
			dest.set(x, y, total);
		}
}
// BlurImageOps.mean(GrayF32, out, radiusX, radiusY, storage)  I:alg/filter/blur/BlurImageOps.java:359-376
//   ConvolveImageMean.horizontal / vertical   I:alg/filter/convolve/ConvolveImageMean.java:55-101
//   interior: running sums, total / divisor   I:alg/filter/convolve/noborder/ImplConvolveMean.java:281-357
//   border: ConvolveNormalized_JustBorder_SB with FactoryKernel.table1D_F32(radius, true)
inline void meanHorizontal(const GrayF32& input, GrayF32& output, int radius) {
	const int kw = radius * 2 + 1;
	std::vector<float> ker(kw, 1.0f / kw);
	if (kw > input.width) { convolveNormalizedHorizontal(ker.data(), kw, radius, input, output); return; }
	convNormBorderHorizontal(ker.data(), kw, radius, input, output);
	const float divisor = kw;
	for (int y = 0; y < input.height; y++) {
		int indexIn = input.startIndex + input.stride * y;
		int indexOut = output.startIndex + output.stride * y + radius;
		float total = 0;
		int indexEnd = indexIn + kw;
		for (; indexIn < indexEnd; indexIn++) total += input.data[indexIn];
		output.data[indexOut++] = total / divisor;
		indexEnd = indexIn + input.width - kw;
		for (; indexIn < indexEnd; indexIn++) {
			total -= input.data[indexIn - kw];
			total += input.data[indexIn];
			output.data[indexOut++] = total / divisor;
		}
	}
}
inline void meanVertical(const GrayF32& input, GrayF32& output, int radius) {
	const int kw = radius * 2 + 1;
	std::vector<float> ker(kw, 1.0f / kw);
	if (kw > input.height) { convolveNormalizedVertical(ker.data(), kw, radius, input, output); return; }
	convNormBorderVertical(ker.data(), kw, radius, input, output);
	const int backStep = kw * input.stride;
	const float divisor = kw;
	const int y0 = radius, y1 = output.height - radius;
	std::vector<float> totals(input.width);
	for (int x = 0; x < input.width; x++) {
		int indexIn = input.startIndex + (y0 - radius) * input.stride + x;
		int indexOut = output.startIndex + output.stride * y0 + x;
		float total = 0;
		int indexEnd = indexIn + input.stride * kw;
		for (; indexIn < indexEnd; indexIn += input.stride) total += input.data[indexIn];
		totals[x] = total;
		output.data[indexOut] = total / divisor;
	}
	for (int y = y0 + 1; y < y1; y++) {
		int indexIn = input.startIndex + (y + radius) * input.stride;
		int indexOut = output.startIndex + y * output.stride;
		for (int x = 0; x < input.width; x++, indexIn++, indexOut++) {
			float total = totals[x] - input.data[indexIn - backStep];
			totals[x] = total += input.data[indexIn];
			output.data[indexOut] = total / divisor;
		}
	}
}
inline void blurMean(const GrayF32& input, GrayF32& output, int radiusX, int radiusY, GrayF32& storage) {
	if (radiusX <= 0 || radiusY <= 0) throw std::invalid_argument("Radius must be > 0");
	meanHorizontal(input, storage, radiusX);
	meanVertical(storage, output, radiusY);
}
// BlurImageOps.median(GrayF32, out, radius) -> ImplMedianSortNaive.process  I:alg/filter/blur/impl/ImplMedianSortNaive.java:97-135:
// window clipped to the image, median = the (count/2)-th order statistic (QuickSelect.select returns a VALUE, so it is pinned)
inline void blurMedian(const GrayF32& input, GrayF32& output, int radius) {
	if (radius <= 0) throw std::invalid_argument("Radius must be > 0");
	std::vector<float> storage((size_t)(2 * radius + 1) * (2 * radius + 1));
	for (int y = 0; y < input.height; y++) {
		int minI = std::max(0, y - radius), maxI = std::min(input.height, y + radius + 1);
		for (int x = 0; x < input.width; x++) {
			int minJ = std::max(0, x - radius), maxJ = std::min(input.width, x + radius + 1);
			int index = 0;
			for (int i = minI; i < maxI; i++)
				for (int j = minJ; j < maxJ; j++) storage[index++] = input.get(j, i);
			std::nth_element(storage.begin(), storage.begin() + index / 2, storage.begin() + index);
			output.set(x, y, storage[index / 2]);
		}
	}
}

// ------------------------------------------------------------------------------------------------
// Gradient corner intensity: ImplSsdCorner_F32 (box window, running sums) with the Shi-Tomasi / Harris scores
//   F:alg/feature/detect/intensity/impl/ImplSsdCornerBox.java:36-51 (border of the intensity image filled with 0)
//   F:alg/feature/detect/intensity/impl/ImplSsdCorner_F32.java:62-127 horizontal(), :133-196 vertical()
//   F:alg/feature/detect/intensity/impl/ShiTomasiCorner_F32.java:33-42, HarrisCorner_F32.java:45-50
// The window sums are RUNNING sums in float (subtract the element leaving, add the one entering), so every value depends on the whole
// history of its row / column: the single-threaded order is restated literally.
// ------------------------------------------------------------------------------------------------
inline float cornerShiTomasi(float totalXX, float totalXY, float totalYY) {
	float left = (totalXX + totalYY) * 0.5f;
	float b = (totalXX - totalYY) * 0.5f;
	float right = (float)std::sqrt((double)(b * b + totalXY * totalXY));
	return left - right;
}
inline float cornerHarris(float kappa, float totalXX, float totalXY, float totalYY) {
	float trace = totalXX + totalYY;
	return (totalXX * totalYY - totalXY * totalXY) - kappa * trace * trace;
}
// kind: 0 Shi-Tomasi, 1 Harris(kappa), 2 the reference test's MockSum (xx + xy + yy)
inline void ssdCornerF32(const GrayF32& derivX, const GrayF32& derivY, int radius, int kind, float kappa, GrayF32& intensity) {
	const int imgWidth = derivX.width, imgHeight = derivX.height;
	if (derivY.width != imgWidth || derivY.height != imgHeight) throw std::invalid_argument("shapes do not match");
	intensity.reshape(imgWidth, imgHeight);
	// ImageMiscOps.fillBorder(intensity, 0, radius); the interior is overwritten below (only where 2*radius < size)
	for (int y = 0; y < imgHeight; y++)
		for (int x = 0; x < imgWidth; x++)
			if (x < radius || x >= imgWidth - radius || y < radius || y >= imgHeight - radius) intensity.set(x, y, 0);
	auto score = [&](float xx, float xy, float yy) { return kind == 0 ? cornerShiTomasi(xx, xy, yy) : kind == 1 ? cornerHarris(kappa, xx, xy, yy) : xx + xy + yy; };
	std::vector<float> hXX((size_t)imgWidth * imgHeight, 0.f), hXY(hXX), hYY(hXX);
	const int windowWidth = radius * 2 + 1, radp1 = radius + 1;
	if (windowWidth > imgWidth || windowWidth > imgHeight) throw std::invalid_argument("window larger than the image");
	for (int row = 0; row < imgHeight; row++) {
		int pix = row * imgWidth;
		int end = pix + windowWidth;
		float totalXX = 0, totalXY = 0, totalYY = 0;
		int indexX = derivX.startIndex + row * derivX.stride;
		int indexY = derivY.startIndex + row * derivY.stride;
		for (; pix < end; pix++) {
			float dx = derivX.data[indexX++], dy = derivY.data[indexY++];
			totalXX += dx * dx; totalXY += dx * dy; totalYY += dy * dy;
		}
		hXX[pix - radp1] = totalXX; hXY[pix - radp1] = totalXY; hYY[pix - radp1] = totalYY;
		end = row * imgWidth + imgWidth;
		for (; pix < end; pix++, indexX++, indexY++) {
			float dx = derivX.data[indexX - windowWidth], dy = derivY.data[indexY - windowWidth];
			totalXX -= dx * dx; totalXY -= dx * dy; totalYY -= dy * dy;
			dx = derivX.data[indexX]; dy = derivY.data[indexY];
			totalXX += dx * dx; totalXY += dx * dy; totalYY += dy * dy;
			hXX[pix - radius] = totalXX; hXY[pix - radius] = totalXY; hYY[pix - radius] = totalYY;
		}
	}
	const int kernelWidth = windowWidth, startX = radius, endX = imgWidth - radius, backStep = kernelWidth * imgWidth;
	const int y0 = radius, y1 = imgHeight - radius;
	std::vector<float> tempXX(imgWidth), tempXY(imgWidth), tempYY(imgWidth);
	float* inten = intensity.data + intensity.startIndex;   // reshape() above made it dense
	for (int x = startX; x < endX; x++) {
		int srcIndex = x + (y0 - radius) * imgWidth;
		int destIndex = imgWidth * y0 + x;
		float totalXX = 0, totalXY = 0, totalYY = 0;
		int indexEnd = srcIndex + imgWidth * kernelWidth;
		for (; srcIndex < indexEnd; srcIndex += imgWidth) { totalXX += hXX[srcIndex]; totalXY += hXY[srcIndex]; totalYY += hYY[srcIndex]; }
		tempXX[x] = totalXX; tempXY[x] = totalXY; tempYY[x] = totalYY;
		inten[destIndex] = score(totalXX, totalXY, totalYY);
	}
	for (int y = y0 + 1; y < y1; y++) {
		int srcIndex = (y + radius) * imgWidth + startX;
		int destIndex = y * imgWidth + startX;
		for (int x = startX; x < endX; x++, srcIndex++, destIndex++) {
			float totalXX = tempXX[x] - hXX[srcIndex - backStep];
			tempXX[x] = totalXX += hXX[srcIndex];
			float totalXY = tempXY[x] - hXY[srcIndex - backStep];
			tempXY[x] = totalXY += hXY[srcIndex];
			float totalYY = tempYY[x] - hYY[srcIndex - backStep];
			tempYY[x] = totalYY += hYY[srcIndex];
			inten[destIndex] = score(totalXX, totalXY, totalYY);
		}
	}
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
