package boofcv.hip;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;

import boofcv.abst.feature.describe.ConfigSurfDescribe;
import boofcv.abst.feature.detect.interest.ConfigFastHessian;
import boofcv.abst.feature.orientation.ConfigAverageIntegral;
import boofcv.abst.feature.orientation.ConfigSlidingIntegral;

/**
 * One bhip_ctx per host thread per device (include/boofhip.h: "one bhip_ctx per host thread"), plus the packing of the reference's
 * Config* classes into the C structs of the ABI (same field order as bhip_fh_cfg / bhip_surf_cfg / bhip_ori_cfg).
 */
public final class BoofHipContext {
	private static final ThreadLocal<long[]> CTX = new ThreadLocal<>();
	public static volatile int device = 0;

	/** The calling thread's context handle (created on first use; fails loudly without a GPU -- there is no CPU fallback in the library). */
	public static long get() {
		long[] h = CTX.get();
		if (h == null) {
			h = new long[1];
			int status = BoofHip.ctxCreate(device, h);
			if (status != 0 || h[0] == 0) throw new RuntimeException("boofhip: bhip_ctx_create(" + device + ") failed with status " + status);
			CTX.set(h);
		}
		return h[0];
	}

	/** A context of its own for one provider object (ADVICE r2: a provider built on one thread and used on another must not share the
	 *  constructing thread's context): created here, destroyed by the provider's close().  The native library tolerates any destroy order
	 *  (include/boofhip.h, "handles may be destroyed in any order"). */
	public static long create() {
		long[] h = new long[1];
		int status = BoofHip.ctxCreate(device, h);
		if (status != 0 || h[0] == 0) throw new RuntimeException("boofhip: bhip_ctx_create(" + device + ") failed with status " + status);
		return h[0];
	}

	/** Releases the calling thread's shared context (the BOverride hooks use it); call from a worker thread before it exits. */
	public static void closeThreadContext() {
		long[] h = CTX.get();
		if (h != null) { BoofHip.ctxDestroy(h[0]); CTX.remove(); }
	}

	private static ByteBuffer struct(int bytes) { return ByteBuffer.allocateDirect(bytes).order(ByteOrder.nativeOrder()); }

	/** bhip_fh_cfg {float detectThreshold; int extractRadius, maxFeaturesPerScale, initialSampleSize, initialSize, numberScalesPerOctave,
	 *  numberOfOctaves, scaleStepSize;} -- ConfigFastHessian.java:33-70; null = reference defaults. */
	public static ByteBuffer pack(ConfigFastHessian c) {
		if (c == null) return null;
		ByteBuffer b = struct(32);
		b.putFloat(c.detectThreshold).putInt(c.extractRadius).putInt(c.maxFeaturesPerScale).putInt(c.initialSampleSize).putInt(c.initialSize)
				.putInt(c.numberScalesPerOctave).putInt(c.numberOfOctaves).putInt(c.scaleStepSize);
		return b;
	}

	/** bhip_surf_cfg {int widthLargeGrid, widthSubRegion, widthSample; double weightSigma; int overLap; double sigmaLargeGrid, sigmaSubRegion;}
	 *  with the C compiler's natural alignment (offsets 0,4,8,16,24,32,40; 48 bytes). */
	public static ByteBuffer pack(ConfigSurfDescribe c) {
		if (c == null) return null;
		if (c.useHaar) throw new RuntimeException("useHaar is not implemented on the GPU");   // BHIP_ERR_UNSUPPORTED: caller keeps the Java path
		ByteBuffer b = struct(48);
		b.putInt(0, c.widthLargeGrid).putInt(4, c.widthSubRegion).putInt(8, c.widthSample);
		if (c instanceof ConfigSurfDescribe.Speed) b.putDouble(16, ((ConfigSurfDescribe.Speed)c).weightSigma); else b.putDouble(16, 4.5);
		if (c instanceof ConfigSurfDescribe.Stability) {
			ConfigSurfDescribe.Stability s = (ConfigSurfDescribe.Stability)c;
			b.putInt(24, s.overLap).putDouble(32, s.sigmaLargeGrid).putDouble(40, s.sigmaSubRegion);
		} else {
			b.putInt(24, 2).putDouble(32, 2.5).putDouble(40, 2.5);
		}
		return b;
	}

	/** bhip_ori_cfg {double objectRadiusToScale, samplePeriod, windowSize; int radius; double weightSigma; int sampleWidth;} (48 bytes) */
	public static ByteBuffer pack(ConfigSlidingIntegral c) {
		if (c == null) return null;
		ByteBuffer b = struct(48);
		b.putDouble(0, c.objectRadiusToScale).putDouble(8, c.samplePeriod).putDouble(16, c.windowSize).putInt(24, c.radius).putDouble(32, c.weightSigma)
				.putInt(40, c.sampleWidth);
		return b;
	}

	public static ByteBuffer pack(ConfigAverageIntegral c) {
		if (c == null) return null;
		ByteBuffer b = struct(48);
		b.putDouble(0, c.objectRadiusToScale).putDouble(8, c.samplePeriod).putDouble(16, 0).putInt(24, c.radius).putDouble(32, c.weightSigma)
				.putInt(40, c.sampleWidth);
		return b;
	}
}
