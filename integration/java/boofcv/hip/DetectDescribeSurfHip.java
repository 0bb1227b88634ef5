package boofcv.hip;

import java.nio.ByteBuffer;

import org.ddogleg.struct.FastQueue;

import boofcv.BoofDefaults;
import boofcv.abst.feature.detdesc.DetectDescribePoint;
import boofcv.struct.feature.BrightFeature;
import boofcv.struct.feature.SurfFeatureQueue;
import boofcv.struct.image.GrayF32;
import boofcv.struct.image.GrayU8;
import boofcv.struct.image.ImageGray;
import georegression.struct.point.Point2D_F64;

/** DetectDescribePoint<T,BrightFeature> over a bhip_surf object (DetectDescribePoint.java:32-46, WrapDetectDescribeSurf.java:93-159) for
 *  T = GrayF32 or GrayU8 (GrayU8: GrayS32 integral images, bhip_surf_detect_u8): results are recycled by the next detect(), instances are
 *  not thread safe -- both as in the reference.  Every instance owns its bhip_ctx (one stream, one set of scratch buffers), so an object
 *  built on one thread may be used on another, as long as one thread uses it at a time.  UNCOMPILED SOURCE (no JDK in the build image). */
public class DetectDescribeSurfHip<T extends ImageGray<T>> implements DetectDescribePoint<T, BrightFeature>, AutoCloseable {
	private final long ctx = BoofHipContext.create();
	private final long surf;
	private final int dof;
	private int n;
	private double[] xys = new double[0], angle = new double[0], desc = new double[0];
	private byte[] white = new byte[0];
	private final FastQueue<BrightFeature> features;
	private final FastQueue<Point2D_F64> locations = new FastQueue<>(Point2D_F64.class, true);
	private final int[] tmp = new int[1];
	private boolean closed;

	DetectDescribeSurfHip(ByteBuffer fh, ByteBuffer surfCfg, ByteBuffer ori, boolean stable) {
		long[] h = new long[1];
		int status = BoofHip.surfCreate(ctx, fh, surfCfg, ori, stable ? 1 : 0, h);
		if (status != 0) { String msg = BoofHip.lastError(ctx); BoofHip.ctxDestroy(ctx); throw new RuntimeException("boofhip: " + msg + " (" + status + ")"); }
		surf = h[0];
		dof = BoofHip.surfDof(surf);
		features = new SurfFeatureQueue(dof);
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

	/** copies image `image` of the last detect into the per-image getters (image 0 after detect()) */
	public void readImage(int image) {
		BoofHip.check(ctx, BoofHip.surfCount(surf, image, tmp));
		n = tmp[0];
		if (xys.length < 3*n) { xys = new double[3*n]; angle = new double[n]; white = new byte[n]; desc = new double[dof*n]; }
		if (n > 0) BoofHip.check(ctx, BoofHip.surfFetch(surf, image, xys, angle, white, desc));
		features.reset(); locations.reset();
		for (int i = 0; i < n; i++) {
			BrightFeature f = features.grow();                    // FastQueue<BrightFeature> <-> contiguous double[n*dof] (SURVEY 8a row a16)
			System.arraycopy(desc, dof*i, f.value, 0, dof);
			f.white = white[i] != 0;
			locations.grow().set(xys[3*i], xys[3*i + 1]);
		}
	}

	@Override public int getNumberOfFeatures() { return n; }
	@Override public Point2D_F64 getLocation(int featureIndex) { return locations.get(featureIndex); }
	@Override public double getRadius(int featureIndex) { return xys[3*featureIndex + 2]*BoofDefaults.SURF_SCALE_TO_RADIUS; }
	@Override public double getOrientation(int featureIndex) { return angle[featureIndex]; }
	@Override public BrightFeature getDescription(int index) { return features.get(index); }
	@Override public BrightFeature createDescription() { return new BrightFeature(dof); }
	@Override public Class<BrightFeature> getDescriptionType() { return BrightFeature.class; }
	@Override public boolean hasScale() { return true; }
	@Override public boolean hasOrientation() { return true; }

	// ---- batch-level calls (no counterpart in the reference interface; what a provider uses when it processes a list of frames) ----
	private int batch;
	private int[] starts = new int[1];

	/** Detect + describe a list of GrayF32 frames of one shape in ONE native call (bhip_surf_detect_f32 with batch > 1: host batches of
	 *  64 frames or more are uploaded in chunks while the previous chunk is processed).  The per-image getters then refer to image 0
	 *  (readImage(i) selects another); use fetchAll / associateImages for the whole batch. */
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

	/** Every location / orientation / sign / descriptor of the last batch in one set of copies (bhip_surf_fetch_all); image i owns rows
	 *  starts()[i] .. starts()[i+1] of the arrays (xy_scale has 3 values per row, desc dof).  The native call writes total() rows: shorter
	 *  arrays are refused here (the shim pins the arrays and cannot check lengths itself). */
	public void fetchAll(double[] xyScale, double[] angles, byte[] whites, double[] descs) {
		final int total = starts[batch];
		if ((xyScale != null && xyScale.length < 3*total) || (angles != null && angles.length < total) || (whites != null && whites.length < total)
				|| (descs != null && descs.length < (long)dof*total))
			throw new IllegalArgumentException("fetchAll: result arrays are shorter than the " + total + " features of the last batch");
		if (total > 0) BoofHip.check(ctx, BoofHip.surfFetchAll(surf, xyScale, angles, whites, descs));
	}
	public int[] starts() { return starts; }
	public int total() { return starts[batch]; }

	/** Greedy Euclidean-squared association of image srcImage[p] with image dstImage[p] of the last batch on the descriptors that are still
	 *  resident on the device (bhip_assoc_l2_surf; AssociateGreedy.java:65-118 rules).  pairs / fit are indexed like fetchAll's rows; rows
	 *  of images that are no source read -1 / 0. */
	public void associateImages(int[] srcImage, int[] dstImage, double maxError, boolean backwardsValidation, int[] pairs, double[] fit) {
		final int total = starts[batch];
		if (srcImage.length != dstImage.length) throw new IllegalArgumentException("source and destination image lists differ in length");
		if (pairs.length < total || fit.length < total) throw new IllegalArgumentException("associateImages: pairs / fit are shorter than the " + total + " features of the last batch");
		BoofHip.check(ctx, BoofHip.assocL2Surf(surf, srcImage.length, srcImage, dstImage, maxError, backwardsValidation ? 1 : 0, pairs, fit));
	}

	/** Java has no deterministic destructor: call when done (try-with-resources).  Order does not matter to the native side. */
	@Override public void close() {
		if (closed) return;
		closed = true;
		BoofHip.surfDestroy(surf);
		BoofHip.ctxDestroy(ctx);
	}
}
