// Host-side synthesis of the small Gaussian weight tables the orientation / SURF kernels read.
// Restates I:factory/filter/kernel/FactoryKernelGaussian.java (gaussian :120-135, gaussian1D_F32 :218-238,
// gaussian2D_F64 :297-306, sigmaForRadius :388, radiusForSigma :404, gaussianWidth :418-448) and
// I:alg/filter/kernel/KernelMath.java (convolve2D :365-383, normalizeSumToOne :417-452).
// UtilGaussian.computePDF is a ddogleg function (not in the reference tree): exp(-d^2/(2 s^2)) / (s sqrt(2 pi)).
#include "common.h"
#include <cmath>

static double pdf(double sigma, double sample) {
	return std::exp(-sample * sample / (2.0 * sigma * sigma)) / (sigma * std::sqrt(2.0 * M_PI));
}
static double sigmaForRadius(double radius) { return (radius * 2.0 + 1.0) / 5.0; }
static int radiusForSigma(double sigma) { return (int)std::ceil((5.0 * sigma - 1) / 2); }

static std::vector<double> outerNormalized(const std::vector<double>& k1) {
	const size_t w = k1.size();
	std::vector<double> out(w * w);
	size_t idx = 0;
	for (size_t i = 0; i < w; i++)
		for (size_t j = 0; j < w; j++) out[idx++] = k1[i] * k1[j];
	double total = 0;
	for (double v : out) total += v;
	for (double& v : out) v /= total;
	return out;
}

std::vector<double> bhip_gaussian2d_f64(double sigma, int radius) {
	if (radius <= 0) radius = radiusForSigma(sigma);
	else if (sigma <= 0) sigma = sigmaForRadius(radius);
	std::vector<double> k1;
	for (int i = radius; i >= -radius; i--) k1.push_back(pdf(sigma, i));
	return outerNormalized(k1);
}

std::vector<double> bhip_gaussian_width(double sigma, int width) {
	if (sigma <= 0) sigma = sigmaForRadius(width / 2);
	if (width % 2 == 1) {
		int radius = width / 2;
		std::vector<double> k1;
		for (int i = radius; i >= -radius; i--) k1.push_back(pdf(sigma, i));
		return outerNormalized(k1);
	}
	const int r = width / 2 - 1;
	std::vector<double> out((size_t)width * width);
	double sum = 0;
	for (int y = 0; y < width; y++) {
		double dy = (y <= r ? std::abs(y - r) : std::abs(y - r - 1)) + 0.5;
		for (int x = 0; x < width; x++) {
			double dx = (x <= r ? std::abs(x - r) : std::abs(x - r - 1)) + 0.5;
			double val = pdf(sigma, std::sqrt(dx * dx + dy * dy));
			out[(size_t)y * width + x] = val;
			sum += val;
		}
	}
	for (double& v : out) v /= sum;
	return out;
}

std::vector<float> bhip_gaussian1d_f32(double sigma, int radius) {
	if (radius <= 0) radius = radiusForSigma(sigma);
	else if (sigma <= 0) sigma = sigmaForRadius(radius);
	std::vector<float> k;
	for (int i = radius; i >= -radius; i--) k.push_back((float)pdf(sigma, i));
	float total = 0;
	for (float v : k) total += v;
	for (float& v : k) v /= total;
	return k;
}
