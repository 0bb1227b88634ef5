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

	/** the native object is released with the owner (Java has no deterministic destructor: call when done) */
	public void close() { BoofHip.surfDestroy(surf); }
}
