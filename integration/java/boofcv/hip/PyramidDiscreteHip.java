package boofcv.hip;

import boofcv.struct.convolve.Kernel1D_F32;
import boofcv.struct.image.GrayF32;
import boofcv.struct.image.ImageType;
import boofcv.struct.pyramid.PyramidDiscrete;

/** PyramidDiscrete<GrayF32> with PyramidDiscreteSampleBlur's semantics (main/boofcv-ip/.../alg/transform/pyramid/PyramidDiscreteSampleBlur.java:
 *  48-141; made by FactoryPyramid.discreteGaussian, .../factory/transform/pyramid/FactoryPyramid.java:53-61): process() runs every layer's
 *  normalised down-convolution pair on the device in one native call (bhip_pyramid_f32) and unpacks the result into the GrayF32 layers
 *  getLayer(i) returns.  The 1-D kernel is built on the Java side (FactoryKernelGaussian.gaussian) and passed in.  UNCOMPILED SOURCE. */
public class PyramidDiscreteHip extends PyramidDiscrete<GrayF32> implements AutoCloseable {
	private final long ctx = BoofHipContext.create();
	private final Kernel1D_F32 kernel;
	private final double[] sigmas;
	private float[] packed = new float[0];
	private int[] dims;
	private long[] offsets;
	private boolean closed;

	public PyramidDiscreteHip(Kernel1D_F32 kernel, double sigma, boolean saveOriginalReference, int... scaleFactors) {
		super(ImageType.single(GrayF32.class), saveOriginalReference, scaleFactors);
		this.kernel = kernel;
		sigmas = new double[scaleFactors.length];          // PyramidDiscreteSampleBlur.java:75-85
		for (int i = 1; i < sigmas.length; i++) {
			double prev = sigmas[i - 1], applied = sigma*scaleFactors[i - 1];
			sigmas[i] = Math.sqrt(prev*prev + applied*applied);
		}
	}

	@Override public void process(GrayF32 input) {
		super.initialize(input.width, input.height);
		final int L = getNumLayers();
		if (dims == null || dims.length != 2*L) { dims = new int[2*L]; offsets = new long[L]; }
		long[] total = new long[1];
		int status = BoofHip.pyramidLayout(input.width, input.height, scale, L, dims, offsets, total);
		if (status != 0) throw new IllegalArgumentException("boofhip: bad pyramid scales");
		if (packed.length < total[0]) packed = new float[(int)total[0]];
		BoofHip.check(ctx, BoofHip.pyramidF32(ctx, kernel.data, kernel.width, scale, L, input.data, input.startIndex, input.stride, input.width, input.height, packed));
		for (int i = 0; i < L; i++) {
			if (i == 0 && scale[0] == 1 && isSaveOriginalReference()) { setFirstLayer(input); continue; }
			GrayF32 layer = getLayer(i);
			final int w = dims[2*i], h = dims[2*i + 1];
			for (int y = 0; y < h; y++) System.arraycopy(packed, (int)offsets[i] + y*w, layer.data, layer.startIndex + y*layer.stride, w);
		}
	}

	@Override public double getSampleOffset(int layer) { return 0; }
	@Override public double getSigma(int layer) { return sigmas[layer]; }

	@Override public void close() { if (!closed) { closed = true; BoofHip.ctxDestroy(ctx); } }
}
