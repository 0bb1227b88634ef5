package boofcv.hip;

import java.nio.ByteBuffer;

import org.ddogleg.struct.FastQueue;

import boofcv.BoofDefaults;
import boofcv.abst.feature.detdesc.DetectDescribePoint;
import boofcv.struct.feature.BrightFeature;
import boofcv.struct.feature.SurfFeatureQueue;
import boofcv.struct.image.GrayF32;
import georegression.struct.point.Point2D_F64;

/** DetectDescribePoint<GrayF32,BrightFeature> over a bhip_surf object (DetectDescribePoint.java:32-46, WrapDetectDescribeSurf.java:93-159): results
 *  are recycled by the next detect(), instances are not thread safe -- both as in the reference. */
public class DetectDescribeSurfHip implements DetectDescribePoint<GrayF32, BrightFeature> {
	private final long ctx = BoofHipContext.get();
	private final long surf;
	private int n;
	private double[] xys = new double[0], angle = new double[0], desc = new double[0];
	private byte[] white = new byte[0];
	private final FastQueue<BrightFeature> features = new SurfFeatureQueue(64);
	private final FastQueue<Point2D_F64> locations = new FastQueue<>(Point2D_F64.class, true);
	private final int[] tmp = new int[1];

	DetectDescribeSurfHip(ByteBuffer fh, ByteBuffer surfCfg, ByteBuffer ori, boolean stable) {
		long[] h = new long[1];
		BoofHip.check(ctx, BoofHip.surfCreate(ctx, fh, surfCfg, ori, stable ? 1 : 0, h));
		surf = h[0];
	}

	@Override public void detect(GrayF32 input) {
		BoofHip.check(ctx, BoofHip.surfDetectF32(surf, new float[][]{input.data}, new int[]{input.startIndex}, new int[]{input.stride}, input.width, input.height, 1));
		BoofHip.check(ctx, BoofHip.surfCount(surf, 0, tmp));
		n = tmp[0];
		if (xys.length < 3*n) { xys = new double[3*n]; angle = new double[n]; white = new byte[n]; desc = new double[64*n]; }
		if (n > 0) BoofHip.check(ctx, BoofHip.surfFetch(surf, 0, xys, angle, white, desc));
		features.reset(); locations.reset();
		for (int i = 0; i < n; i++) {
			BrightFeature f = features.grow();                    // FastQueue<BrightFeature> <-> contiguous double[n*64] (SURVEY 8a row a16)
			System.arraycopy(desc, 64*i, f.value, 0, 64);
			f.white = white[i] != 0;
			locations.grow().set(xys[3*i], xys[3*i + 1]);
		}
	}

	@Override public int getNumberOfFeatures() { return n; }
	@Override public Point2D_F64 getLocation(int featureIndex) { return locations.get(featureIndex); }
	@Override public double getRadius(int featureIndex) { return xys[3*featureIndex + 2]*BoofDefaults.SURF_SCALE_TO_RADIUS; }
	@Override public double getOrientation(int featureIndex) { return angle[featureIndex]; }
	@Override public BrightFeature getDescription(int index) { return features.get(index); }
	@Override public BrightFeature createDescription() { return new BrightFeature(64); }
	@Override public Class<BrightFeature> getDescriptionType() { return BrightFeature.class; }
	@Override public boolean hasScale() { return true; }
	@Override public boolean hasOrientation() { return true; }

	// ---- batch-level calls (no counterpart in the reference interface; what a provider uses when it processes a list of frames) ----
	private int batch;
	private int[] starts = new int[1];

	/** Detect + describe a list of frames of one shape in ONE native call (bhip_surf_detect_f32 with batch > 1: host batches of 64 frames
	 *  or more are uploaded in chunks while the previous chunk is processed).  The per-image getters then refer to image 0; use
	 *  fetchAll / associateImages for the whole batch. */
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
	}

	/** Every location / orientation / sign / descriptor of the last batch in one set of copies (bhip_surf_fetch_all); image i owns rows
	 *  starts()[i] .. starts()[i+1] of the returned arrays (xy_scale has 3 values per row, desc 64). */
	public void fetchAll(double[] xyScale, double[] angles, byte[] whites, double[] descs) {
		if (starts[batch] > 0) BoofHip.check(ctx, BoofHip.surfFetchAll(surf, xyScale, angles, whites, descs));
	}
	public int[] starts() { return starts; }

	/** Greedy Euclidean-squared association of image srcImage[p] with image dstImage[p] of the last batch on the descriptors that are still
	 *  resident on the device (bhip_assoc_l2_surf; AssociateGreedy.java:65-118 rules).  pairs / fit are indexed like fetchAll's rows. */
	public void associateImages(int[] srcImage, int[] dstImage, double maxError, boolean backwardsValidation, int[] pairs, double[] fit) {
		BoofHip.check(ctx, BoofHip.assocL2Surf(surf, srcImage.length, srcImage, dstImage, maxError, backwardsValidation ? 1 : 0, pairs, fit));
	}

	/** the native object is released with the owner (Java has no deterministic destructor: call when done) */
	public void close() { BoofHip.surfDestroy(surf); }
}
