package boofcv.hip;

import boofcv.abst.feature.describe.DescribeRegionPoint;
import boofcv.alg.feature.describe.brief.BinaryCompareDefinition_I32;
import boofcv.struct.feature.TupleDesc_B;
import boofcv.struct.image.GrayF32;
import boofcv.struct.image.ImageType;

/** DescribeRegionPoint<GrayF32,TupleDesc_B> = WrapDescribeBrief over DescribePointBrief (main/boofcv-feature/.../abst/feature/describe/
 *  WrapDescribeBrief.java:30-86, .../alg/feature/describe/DescribePointBrief.java:71-89) through bhip_brief_f32: the fixed BRIEF samples the
 *  frame handed to setImage (the blurred copy the reference makes is never read), ignores orientation and radius, always succeeds.
 *  processAll() describes a whole point list in one native call.  UNCOMPILED SOURCE. */
public class DescribeBriefHip implements DescribeRegionPoint<GrayF32, TupleDesc_B>, AutoCloseable {
	private final long ctx = BoofHipContext.create();
	private final BriefDefinitionHip def;
	private GrayF32 image;
	private final double[] one = new double[2];
	private boolean closed;

	public DescribeBriefHip(BinaryCompareDefinition_I32 definition) { def = new BriefDefinitionHip(definition); }

	@Override public void setImage(GrayF32 image) { this.image = image; }

	@Override public boolean process(double x, double y, double orientation, double radius, TupleDesc_B description) {
		one[0] = x; one[1] = y;
		BoofHip.check(ctx, BoofHip.briefF32(ctx, image.data, image.startIndex, image.stride, image.width, image.height, def.radius, def.numPoints, def.samplePoints,
				def.compare, one, 1, description.data));
		return true;
	}

	/** xy = (x0,y0,x1,y1,...): n points; words = n * ceil(numPoints/32) ints */
	public void processAll(double[] xy, int n, int[] words) {
		if (xy.length < 2*n || words.length < (long)n*def.words()) throw new IllegalArgumentException("processAll: arrays too short for " + n + " points");
		BoofHip.check(ctx, BoofHip.briefF32(ctx, image.data, image.startIndex, image.stride, image.width, image.height, def.radius, def.numPoints, def.samplePoints,
				def.compare, xy, n, words));
	}

	@Override public boolean requiresRadius() { return false; }
	@Override public boolean requiresOrientation() { return false; }
	@Override public ImageType<GrayF32> getImageType() { return ImageType.single(GrayF32.class); }
	@Override public double getCanonicalWidth() { return def.radius*2 + 1; }
	@Override public TupleDesc_B createDescription() { return new TupleDesc_B(def.numPoints); }
	@Override public Class<TupleDesc_B> getDescriptionType() { return TupleDesc_B.class; }

	@Override public void close() { if (!closed) { closed = true; BoofHip.ctxDestroy(ctx); } }
}
