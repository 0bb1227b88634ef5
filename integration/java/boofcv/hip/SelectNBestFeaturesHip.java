package boofcv.hip;

import boofcv.alg.feature.detect.extract.SelectNBestFeatures;
import boofcv.struct.QueueCorner;
import boofcv.struct.image.GrayF32;
import georegression.struct.point.Point2D_I16;

/** SelectNBestFeatures (main/boofcv-feature/.../alg/feature/detect/extract/SelectNBestFeatures.java:32-98) with process() on the device
 *  (bhip_select_nbest_f32).  The kept SET is exact; the ORDER of the kept corners follows the library's restatement of ddogleg's
 *  QuickSelect.selectIndex and is unpinned against the jar (DESIGN.md section 2).  UNCOMPILED SOURCE. */
public class SelectNBestFeaturesHip extends SelectNBestFeatures implements AutoCloseable {
	private final long ctx = BoofHipContext.create();
	private final QueueCorner best = new QueueCorner(10);
	private int targetN;
	private short[] in = new short[0], out = new short[0];
	private final int[] outN = new int[1];
	private boolean closed;

	public SelectNBestFeaturesHip(int N) { super(N); targetN = N; }

	@Override public void setN(int N) { super.setN(N); targetN = N; }

	@Override public void process(GrayF32 intensityImage, QueueCorner origCorners, boolean positive) {
		final int n = origCorners.size;
		if (in.length < 2*n) { in = new short[2*n]; out = new short[2*n]; }
		for (int i = 0; i < n; i++) { Point2D_I16 p = origCorners.data[i]; in[2*i] = p.x; in[2*i + 1] = p.y; }
		BoofHip.check(ctx, BoofHip.selectNbestF32(ctx, intensityImage.data, intensityImage.startIndex, intensityImage.stride, intensityImage.width, intensityImage.height,
				in, n, targetN, positive ? 1 : 0, out, outN));
		best.reset();
		for (int i = 0; i < outN[0]; i++) best.add(out[2*i], out[2*i + 1]);
	}

	@Override public QueueCorner getBestCorners() { return best; }

	@Override public void close() { if (!closed) { closed = true; BoofHip.ctxDestroy(ctx); } }
}
