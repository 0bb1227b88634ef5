package boofcv.hip;

import javax.annotation.Nullable;

import boofcv.abst.feature.describe.ConfigSurfDescribe;
import boofcv.abst.feature.detdesc.DetectDescribePoint;
import boofcv.abst.feature.detect.interest.ConfigFastHessian;
import boofcv.abst.feature.orientation.ConfigAverageIntegral;
import boofcv.abst.feature.orientation.ConfigSlidingIntegral;
import boofcv.struct.feature.BrightFeature;
import boofcv.struct.image.GrayF32;

/** Same signatures as FactoryDetectDescribe.surfFast / surfStable (main/boofcv-feature/.../factory/feature/detdesc/FactoryDetectDescribe.java:118-135,
 *  209-226) for GrayF32, returning the same interface type, backed by libboofhip.so. */
public class FactoryDetectDescribeHip {
	public static DetectDescribePoint<GrayF32, BrightFeature>
	surfFast(@Nullable ConfigFastHessian configDetector, @Nullable ConfigSurfDescribe.Speed configDesc, @Nullable ConfigAverageIntegral configOrientation, Class<GrayF32> imageType) {
		return new DetectDescribeSurfHip(BoofHipContext.pack(configDetector), BoofHipContext.pack(configDesc), BoofHipContext.pack(configOrientation), false);
	}

	public static DetectDescribePoint<GrayF32, BrightFeature>
	surfStable(@Nullable ConfigFastHessian configDetector, @Nullable ConfigSurfDescribe.Stability configDescribe, @Nullable ConfigSlidingIntegral configOrientation, Class<GrayF32> imageType) {
		return new DetectDescribeSurfHip(BoofHipContext.pack(configDetector), BoofHipContext.pack(configDescribe), BoofHipContext.pack(configOrientation), true);
	}
}
