package boofcv.hip;

import java.nio.ByteBuffer;

import org.ddogleg.struct.FastQueue;

import boofcv.BoofDefaults;
import boofcv.abst.feature.detdesc.DetectDescribePoint;
import boofcv.alg.feature.describe.brief.BinaryCompareDefinition_I32;
import boofcv.struct.feature.TupleDesc_B;
import boofcv.struct.image.GrayF32;
import boofcv.struct.image.GrayU8;
import boofcv.struct.image.ImageGray;
import georegression.struct.point.Point2D_F64;

/** DetectDescribePoint<T,TupleDesc_B> = DetectDescribeFusion(FactoryInterestPoint.fastHessian(config), null, FactoryDescribeRegionPoint.brief(...))
 *  (main/boofcv-feature/.../abst/feature/detdesc/DetectDescribeFusion.java:45-165, .../factory/feature/detdesc/FactoryDetectDescribe.java:279-284)
 *  over a bhip_surf created with bhip_surf_create_brief: integral image, Fast-Hessian, BRIEF words and -- through associateImages -- the
 *  Hamming association all stay on the device.  Every detected point is described (WrapDescribeBrief.process always returns true), in
 *  detector order; orientation is 0 (WrapFHtoInterestPoint.java:77-79), radius = scale * 2.  T = GrayF32 or GrayU8.  UNCOMPILED SOURCE. */
public class DetectDescribeFusionHip<T extends ImageGray<T>> implements DetectDescribePoint<T, TupleDesc_B>, AutoCloseable {
	private final long ctx = BoofHipContext.create();
	private final long surf;
	private final BriefDefinitionHip def;
	private int n;
	private double[] xys = new double[0];
	private int[] words = new int[0];
	private final FastQueue<TupleDesc_B> features;
	private final FastQueue<Point2D_F64> locations = new FastQueue<>(Point2D_F64.class, true);
	private final int[] tmp = new int[1];
	private int batch;
	private int[] starts = new int[1];
	private boolean closed;

	DetectDescribeFusionHip(ByteBuffer fh, BinaryCompareDefinition_I32 definition) {
		def = new BriefDefinitionHip(definition);
		long[] h = new long[1];
		int status = BoofHip.surfCreateBrief(ctx, fh, def.radius, def.numPoints, def.samplePoints, def.compare, h);
		if (status != 0) { String msg = BoofHip.lastError(ctx); BoofHip.ctxDestroy(ctx); throw new RuntimeException("boofhip: " + msg + " (" + status + ")"); }
		surf = h[0];
		final int numBits = def.numPoints;
		features = new FastQueue<TupleDesc_B>(TupleDesc_B.class, true) {
			@Override protected TupleDesc_B createInstance() { return new TupleDesc_B(numBits); }
		};
	}

	@Override public void detect(T input) {
		if (input instanceof GrayF32) {
			GrayF32 in = (GrayF32)input;
			BoofHip.check(ctx, BoofHip.surfDetectF32(surf, new float[][]{in.data}, new int[]{in.startIndex}, new int[]{in.stride}, in.width, in.height, 1));
		} else if (input instanceof GrayU8) {
			GrayU8 in = (GrayU8)input;
			BoofHip.check(ctx, BoofHip.surfDetectU8(surf, new byte[][]{in.data}, new int[]{in.startIndex}, new int[]{in.stride}, in.width, in.height, 1));
		} else {
			throw new IllegalArgumentException("Image type not supported");
		}
		batch = 1;
		starts = new int[]{0, 0};
		readImage(0);
		starts[1] = n;
	}

	/** copies image `image` of the last detect into the per-image getters */
	public void readImage(int image) {
		final int w = def.words();
		BoofHip.check(ctx, BoofHip.surfCount(surf, image, tmp));
		n = tmp[0];
		if (xys.length < 3*n) { xys = new double[3*n]; words = new int[w*n]; }
		if (n > 0) {
			BoofHip.check(ctx, BoofHip.surfFetch(surf, image, xys, null, null, null));
			BoofHip.check(ctx, BoofHip.surfFetchBrief(surf, image, words));
		}
		features.reset(); locations.reset();
		for (int i = 0; i < n; i++) {
			System.arraycopy(words, w*i, features.grow().data, 0, w);
			locations.grow().set(xys[3*i], xys[3*i + 1]);
		}
	}

	@Override public int getNumberOfFeatures() { return n; }
	@Override public Point2D_F64 getLocation(int featureIndex) { return locations.get(featureIndex); }
	@Override public double getRadius(int featureIndex) { return xys[3*featureIndex + 2]*BoofDefaults.SURF_SCALE_TO_RADIUS; }
	@Override public double getOrientation(int featureIndex) { return 0; }
	@Override public TupleDesc_B getDescription(int index) { return features.get(index); }
	@Override public TupleDesc_B createDescription() { return new TupleDesc_B(def.numPoints); }
	@Override public Class<TupleDesc_B> getDescriptionType() { return TupleDesc_B.class; }
	@Override public boolean hasScale() { return true; }
	@Override public boolean hasOrientation() { return false; }   // orientation == null -> detector.hasOrientation()

	// ---- batch-level calls ----
	public void detectBatch(java.util.List<GrayF32> frames) {
		if (frames.isEmpty()) throw new IllegalArgumentException("empty batch");
		final int w = frames.get(0).width, h = frames.get(0).height;
		float[][] data = new float[frames.size()][];
		int[] start = new int[frames.size()], stride = new int[frames.size()];
		for (int i = 0; i < frames.size(); i++) {
			GrayF32 f = frames.get(i);
			if (f.width != w || f.height != h) throw new IllegalArgumentException("all images of a batch must have the same shape");
			data[i] = f.data; start[i] = f.startIndex; stride[i] = f.stride;
		}
		BoofHip.check(ctx, BoofHip.surfDetectF32(surf, data, start, stride, w, h, frames.size()));
		batch = frames.size();
		starts = new int[batch + 1];
		for (int i = 0; i < batch; i++) { BoofHip.check(ctx, BoofHip.surfCount(surf, i, tmp)); starts[i + 1] = starts[i] + tmp[0]; }
		readImage(0);
	}
	public int[] starts() { return starts; }
	public int total() { return starts[batch]; }

	/** every location (3 values per row) and every word list (ceil(numPoints/32) ints per row) of the last batch */
	public void fetchAll(double[] xyScale, int[] allWords) {
		final int total = starts[batch];
		if ((xyScale != null && xyScale.length < 3*total) || (allWords != null && allWords.length < (long)def.words()*total))
			throw new IllegalArgumentException("fetchAll: result arrays are shorter than the " + total + " features of the last batch");
		if (total == 0) return;
		if (xyScale != null) BoofHip.check(ctx, BoofHip.surfFetchAll(surf, xyScale, null, null, null));
		if (allWords != null) BoofHip.check(ctx, BoofHip.surfFetchBrief(surf, -1, allWords));
	}

	/** Greedy Hamming association (ScoreAssociateHamming_B, AssociateGreedy.java:65-118 rules) of image srcImage[p] with image dstImage[p] of
	 *  the last batch on the words still resident on the device (bhip_assoc_hamming_surf) */
	public void associateImages(int[] srcImage, int[] dstImage, double maxError, boolean backwardsValidation, int[] pairs, double[] fit) {
		final int total = starts[batch];
		if (srcImage.length != dstImage.length) throw new IllegalArgumentException("source and destination image lists differ in length");
		if (pairs.length < total || fit.length < total) throw new IllegalArgumentException("associateImages: pairs / fit are shorter than the " + total + " features of the last batch");
		BoofHip.check(ctx, BoofHip.assocHammingSurf(surf, srcImage.length, srcImage, dstImage, maxError, backwardsValidation ? 1 : 0, pairs, fit));
	}

	@Override public void close() {
		if (closed) return;
		closed = true;
		BoofHip.surfDestroy(surf);
		BoofHip.ctxDestroy(ctx);
	}
}
