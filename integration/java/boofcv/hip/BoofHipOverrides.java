package boofcv.hip;

import boofcv.abst.feature.detect.extract.ConfigExtract;
import boofcv.abst.feature.detect.extract.NonMaxSuppression;
import boofcv.alg.filter.blur.BOverrideBlurImageOps;
import boofcv.alg.filter.convolve.BOverrideConvolveImage;
import boofcv.alg.filter.convolve.BOverrideConvolveImageNormalized;
import boofcv.alg.filter.convolve.border.ConvolveJustBorder_General_SB;
import boofcv.core.image.border.ImageBorder_F32;
import boofcv.factory.feature.detect.extract.BOverrideFactoryFeatureExtractor;
import boofcv.struct.QueueCorner;
import boofcv.struct.convolve.Kernel1D_F32;
import boofcv.struct.convolve.Kernel2D_F32;
import boofcv.struct.image.GrayF32;

/**
 * Installs every BOverride hook the library implements (I:override/BOverrideManager.java:33-39 lists the hook classes).  A hook that throws
 * RuntimeException means "not handled, run the Java code" (BOverrideConvolveImage.java:53-62), which is what BoofHip.check does for any
 * non-zero status and what the type guards below do for images / kernels other than GrayF32 / Kernel*_F32.
 */
public final class BoofHipOverrides {
	private BoofHipOverrides() {}

	private static RuntimeException declined() { return new RuntimeException("boofhip: not handled"); }

	public static void install() {
		// ConvolveImage.horizontal/vertical/convolve(kernel, input, output, border) = no-border interior (GPU) + ConvolveJustBorder_General_SB (Java strip)
		BOverrideConvolveImage.horizontal = (kernel, input, output, border) -> {
			if (!(kernel instanceof Kernel1D_F32) || !(input instanceof GrayF32) || !(border instanceof ImageBorder_F32)) throw declined();
			Kernel1D_F32 k = (Kernel1D_F32)kernel; GrayF32 in = (GrayF32)input, out = (GrayF32)output;
			long ctx = BoofHipContext.get();
			BoofHip.check(ctx, BoofHip.convHF32(ctx, k.data, k.width, k.offset, in.data, in.startIndex, in.stride, in.width, in.height, out.data, out.startIndex, out.stride));
			((ImageBorder_F32)border).setImage(in);
			ConvolveJustBorder_General_SB.horizontal(k, (ImageBorder_F32)border, out);
		};
		BOverrideConvolveImage.vertical = (kernel, input, output, border) -> {
			if (!(kernel instanceof Kernel1D_F32) || !(input instanceof GrayF32) || !(border instanceof ImageBorder_F32)) throw declined();
			Kernel1D_F32 k = (Kernel1D_F32)kernel; GrayF32 in = (GrayF32)input, out = (GrayF32)output;
			long ctx = BoofHipContext.get();
			BoofHip.check(ctx, BoofHip.convVF32(ctx, k.data, k.width, k.offset, in.data, in.startIndex, in.stride, in.width, in.height, out.data, out.startIndex, out.stride));
			((ImageBorder_F32)border).setImage(in);
			ConvolveJustBorder_General_SB.vertical(k, (ImageBorder_F32)border, out);
		};
		BOverrideConvolveImage.convolve = (kernel, input, output, border) -> {
			if (!(kernel instanceof Kernel2D_F32) || !(input instanceof GrayF32) || !(border instanceof ImageBorder_F32)) throw declined();
			Kernel2D_F32 k = (Kernel2D_F32)kernel; GrayF32 in = (GrayF32)input, out = (GrayF32)output;
			long ctx = BoofHipContext.get();
			BoofHip.check(ctx, BoofHip.conv2dF32(ctx, k.data, k.width, k.offset, in.data, in.startIndex, in.stride, in.width, in.height, out.data, out.startIndex, out.stride));
			((ImageBorder_F32)border).setImage(in);
			ConvolveJustBorder_General_SB.convolve(k, (ImageBorder_F32)border, out);
		};
		// ConvolveImageNormalized.horizontal / vertical (the 2-D normalised form is not implemented on the GPU: left null = Java)
		BOverrideConvolveImageNormalized.horizontal = (kernel, input, output) -> {
			if (!(kernel instanceof Kernel1D_F32) || !(input instanceof GrayF32)) throw declined();
			Kernel1D_F32 k = (Kernel1D_F32)kernel; GrayF32 in = (GrayF32)input, out = (GrayF32)output;
			long ctx = BoofHipContext.get();
			BoofHip.check(ctx, BoofHip.convNormHF32(ctx, k.data, k.width, k.offset, in.data, in.startIndex, in.stride, in.width, in.height, out.data, out.startIndex, out.stride));
		};
		BOverrideConvolveImageNormalized.vertical = (kernel, input, output) -> {
			if (!(kernel instanceof Kernel1D_F32) || !(input instanceof GrayF32)) throw declined();
			Kernel1D_F32 k = (Kernel1D_F32)kernel; GrayF32 in = (GrayF32)input, out = (GrayF32)output;
			long ctx = BoofHipContext.get();
			BoofHip.check(ctx, BoofHip.convNormVF32(ctx, k.data, k.width, k.offset, in.data, in.startIndex, in.stride, in.width, in.height, out.data, out.startIndex, out.stride));
		};
		// BlurImageOps.mean / median / gaussian (BOverrideBlurImageOps.java:36-50)
		BOverrideBlurImageOps.mean = (input, output, radiusX, radiusY, storage) -> {
			if (!(input instanceof GrayF32)) throw declined();
			GrayF32 in = (GrayF32)input, out = (GrayF32)output;
			long ctx = BoofHipContext.get();
			BoofHip.check(ctx, BoofHip.meanF32(ctx, in.data, in.startIndex, in.stride, in.width, in.height, radiusX, radiusY, out.data, out.startIndex, out.stride));
		};
		BOverrideBlurImageOps.median = (input, output, radius) -> {
			if (!(input instanceof GrayF32)) throw declined();
			GrayF32 in = (GrayF32)input, out = (GrayF32)output;
			long ctx = BoofHipContext.get();
			BoofHip.check(ctx, BoofHip.medianF32(ctx, in.data, in.startIndex, in.stride, in.width, in.height, radius, out.data, out.startIndex, out.stride));
		};
		BOverrideBlurImageOps.gaussian = (input, output, sigmaX, radiusX, sigmaY, radiusY, storage) -> {
			if (!(input instanceof GrayF32) || sigmaX != sigmaY || radiusX != radiusY) throw declined();   // the C ABI has the isotropic form
			GrayF32 in = (GrayF32)input, out = (GrayF32)output;
			long ctx = BoofHipContext.get();
			BoofHip.check(ctx, BoofHip.gaussianF32(ctx, in.data, in.startIndex, in.stride, in.width, in.height, sigmaX, radiusX, out.data, out.startIndex, out.stride));
		};
		// FactoryFeatureExtractor.nonmax(ConfigExtract): strict maxima only (relaxed rule / minima / candidate lists stay on the Java path)
		BOverrideFactoryFeatureExtractor.nonmax = (ConfigExtract config) -> {
			config.checkValidity();
			if (!config.useStrictRule || config.detectMinimums || !config.detectMaximums) throw declined();
			return new NonMaxHip(config);
		};
	}

	/** NonMaxSuppression over bhip_nonmax_block_f32 (NonMaxBlock.process with the strict search, block-raster order). */
	static final class NonMaxHip implements NonMaxSuppression {
		private int radius, border; private float threshold;
		private short[] xy = new short[0];
		private final int[] n = new int[1];
		NonMaxHip(ConfigExtract c) { radius = c.radius; border = c.ignoreBorder; threshold = c.threshold; }

		@Override public void process(GrayF32 intensity, QueueCorner candidateMin, QueueCorner candidateMax, QueueCorner foundMin, QueueCorner foundMax) {
			long ctx = BoofHipContext.get();
			int step = radius + 1;
			int cap = Math.max(1, ((intensity.width - 2*border + step - 1)/step)*((intensity.height - 2*border + step - 1)/step));
			if (xy.length < 2*cap) xy = new short[2*cap];
			BoofHip.check(ctx, BoofHip.nonmaxBlockF32(ctx, intensity.data, intensity.startIndex, intensity.stride, intensity.width, intensity.height, radius, threshold,
					border, xy, cap, n));
			foundMax.reset();
			for (int i = 0; i < n[0]; i++) foundMax.add(xy[2*i], xy[2*i + 1]);
		}
		@Override public boolean getUsesCandidates() { return false; }
		@Override public float getThresholdMinimum() { return -threshold; }
		@Override public float getThresholdMaximum() { return threshold; }
		@Override public void setThresholdMinimum(float t) {}
		@Override public void setThresholdMaximum(float t) { threshold = t; }
		@Override public void setIgnoreBorder(int b) { border = b; }
		@Override public int getIgnoreBorder() { return border; }
		@Override public void setSearchRadius(int r) { radius = r; }
		@Override public int getSearchRadius() { return radius; }
		@Override public boolean canDetectMaximums() { return true; }
		@Override public boolean canDetectMinimums() { return false; }
	}
}
