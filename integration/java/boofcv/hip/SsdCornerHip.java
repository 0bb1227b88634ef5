package boofcv.hip;

import boofcv.alg.feature.detect.intensity.GradientCornerIntensity;
import boofcv.struct.image.GrayF32;

/** GradientCornerIntensity<GrayF32>: the unweighted Shi-Tomasi / Harris intensity of FactoryIntensityPointAlg.shiTomasi / harris
 *  (main/boofcv-feature/.../factory/feature/detect/intensity/FactoryIntensityPointAlg.java:91-160 -> .../alg/feature/detect/intensity/impl/
 *  ImplSsdCorner_F32.java:62-196, ShiTomasiCorner_F32.java:33-42, HarrisCorner_F32.java:45-50) through bhip_corner_intensity_f32; handed to the
 *  unchanged GeneralFeatureDetector.  The running box sums keep the reference's single-threaded order (bit-exact).  UNCOMPILED SOURCE. */
public class SsdCornerHip implements GradientCornerIntensity<GrayF32>, AutoCloseable {
	public static final int SHI_TOMASI = 0, HARRIS = 1;
	private final long ctx = BoofHipContext.create();
	private final int kind, radius;
	private final float kappa;
	private boolean closed;

	public SsdCornerHip(int kind, int radius, float kappa) { this.kind = kind; this.radius = radius; this.kappa = kappa; }

	@Override public void process(GrayF32 derivX, GrayF32 derivY, GrayF32 intensity) {
		if (derivX.width != derivY.width || derivX.height != derivY.height || derivX.startIndex != derivY.startIndex || derivX.stride != derivY.stride)
			throw new IllegalArgumentException("derivX and derivY must have the same shape and layout");
		intensity.reshape(derivX.width, derivX.height);
		BoofHip.check(ctx, BoofHip.cornerIntensityF32(ctx, kind, radius, kappa, derivX.data, derivY.data, derivX.startIndex, derivX.stride, derivX.width, derivX.height,
				intensity.data, intensity.startIndex, intensity.stride));
	}

	@Override public int getRadius() { return radius; }
	@Override public int getIgnoreBorder() { return radius; }

	@Override public void close() { if (!closed) { closed = true; BoofHip.ctxDestroy(ctx); } }
}
