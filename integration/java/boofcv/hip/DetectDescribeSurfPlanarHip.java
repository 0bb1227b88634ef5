package boofcv.hip;

import java.nio.ByteBuffer;

import org.ddogleg.struct.FastQueue;

import boofcv.abst.feature.detdesc.DetectDescribePoint;
import boofcv.struct.feature.BrightFeature;
import boofcv.struct.feature.SurfFeatureQueue;
import boofcv.struct.image.GrayF32;
import boofcv.struct.image.Planar;
import georegression.struct.point.Point2D_F64;

/** DetectDescribePoint<Planar<GrayF32>,BrightFeature> = SurfPlanar_to_DetectDescribePoint (main/boofcv-feature/.../abst/feature/detdesc/
 *  SurfPlanar_to_DetectDescribePoint.java:40-132) over bhip_surf_detect_planar_f32: key points from the band average, one 64-value block per
 *  band, normalised as a whole; getRadius(i) is the scale itself (DetectDescribeSurfPlanar.java:110-124).  UNCOMPILED SOURCE. */
public class DetectDescribeSurfPlanarHip implements DetectDescribePoint<Planar<GrayF32>, BrightFeature>, AutoCloseable {
	private final long ctx = BoofHipContext.create();
	private final long surf;
	private final int numBands, dof;
	private int n;
	private double[] xys = new double[0], angle = new double[0], desc = new double[0];
	private byte[] white = new byte[0];
	private final FastQueue<BrightFeature> features;
	private final FastQueue<Point2D_F64> locations = new FastQueue<>(Point2D_F64.class, true);
	private final int[] tmp = new int[1];
	private boolean closed;

	DetectDescribeSurfPlanarHip(ByteBuffer fh, ByteBuffer surfCfg, ByteBuffer ori, boolean stable, int numBands) {
		long[] h = new long[1];
		int status = BoofHip.surfCreate(ctx, fh, surfCfg, ori, stable ? 1 : 0, h);
		if (status != 0) { String msg = BoofHip.lastError(ctx); BoofHip.ctxDestroy(ctx); throw new RuntimeException("boofhip: " + msg + " (" + status + ")"); }
		surf = h[0];
		this.numBands = numBands;
		dof = 64*numBands;
		features = new SurfFeatureQueue(dof);
	}

	@Override public void detect(Planar<GrayF32> input) {
		if (input.getNumBands() != numBands)
			throw new IllegalArgumentException("Unexpected number of bands. Expected " + numBands + " found " + input.getNumBands());
		float[][] bands = new float[numBands][];
		for (int b = 0; b < numBands; b++) {
			GrayF32 band = input.getBand(b);
			if (band.startIndex != input.startIndex || band.stride != input.stride) throw new IllegalArgumentException("bands must share startIndex and stride");
			bands[b] = band.data;
		}
		BoofHip.check(ctx, BoofHip.surfDetectPlanarF32(surf, bands, numBands, input.startIndex, input.stride, input.width, input.height));
		BoofHip.check(ctx, BoofHip.surfCount(surf, 0, tmp));
		n = tmp[0];
		if (xys.length < 3*n) { xys = new double[3*n]; angle = new double[n]; white = new byte[n]; desc = new double[dof*n]; }
		if (n > 0) BoofHip.check(ctx, BoofHip.surfFetch(surf, 0, xys, angle, white, desc));
		features.reset(); locations.reset();
		for (int i = 0; i < n; i++) {
			BrightFeature f = features.grow();
			System.arraycopy(desc, dof*i, f.value, 0, dof);
			f.white = white[i] != 0;
			locations.grow().set(xys[3*i], xys[3*i + 1]);
		}
	}

	@Override public int getNumberOfFeatures() { return n; }
	@Override public Point2D_F64 getLocation(int featureIndex) { return locations.get(featureIndex); }
	@Override public double getRadius(int featureIndex) { return xys[3*featureIndex + 2]; }   // SurfPlanar_to_DetectDescribePoint.java:100-102
	@Override public double getOrientation(int featureIndex) { return angle[featureIndex]; }
	@Override public BrightFeature getDescription(int index) { return features.get(index); }
	@Override public BrightFeature createDescription() { return new BrightFeature(dof); }
	@Override public Class<BrightFeature> getDescriptionType() { return BrightFeature.class; }
	@Override public boolean hasScale() { return true; }
	@Override public boolean hasOrientation() { return true; }

	@Override public void close() {
		if (closed) return;
		closed = true;
		BoofHip.surfDestroy(surf);
		BoofHip.ctxDestroy(ctx);
	}
}
