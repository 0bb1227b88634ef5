package boofcv.hip;

import java.util.Random;

import javax.annotation.Nullable;

import boofcv.abst.feature.describe.ConfigBrief;
import boofcv.abst.feature.describe.ConfigSurfDescribe;
import boofcv.abst.feature.detdesc.DetectDescribePoint;
import boofcv.abst.feature.detect.interest.ConfigFastHessian;
import boofcv.abst.feature.orientation.ConfigAverageIntegral;
import boofcv.abst.feature.orientation.ConfigSlidingIntegral;
import boofcv.alg.feature.describe.brief.BinaryCompareDefinition_I32;
import boofcv.alg.feature.describe.brief.FactoryBriefDefinition;
import boofcv.struct.feature.BrightFeature;
import boofcv.struct.feature.TupleDesc_B;
import boofcv.struct.image.GrayF32;
import boofcv.struct.image.GrayU8;
import boofcv.struct.image.ImageGray;
import boofcv.struct.image.ImageType;
import boofcv.struct.image.Planar;

/** Same signatures as FactoryDetectDescribe (main/boofcv-feature/.../factory/feature/detdesc/FactoryDetectDescribe.java) for what the library
 *  implements, returning the same interface types, backed by libboofhip.so: surfFast / surfStable (:118-135, 209-226) on GrayF32 and GrayU8,
 *  surfColorFast / surfColorStable (:154-176, 246-268) on Planar<GrayF32>, and fuseTogether(fastHessian, null, brief) (:279-284).
 *  UNCOMPILED SOURCE (no JDK in the build image). */
public class FactoryDetectDescribeHip {
	private static void checkGray(Class<?> imageType) {
		if (imageType != GrayF32.class && imageType != GrayU8.class)
			throw new IllegalArgumentException("Image type not supported");   // the reference's own message for types it does not handle
	}

	public static <T extends ImageGray<T>> DetectDescribePoint<T, BrightFeature>
	surfFast(@Nullable ConfigFastHessian configDetector, @Nullable ConfigSurfDescribe.Speed configDesc, @Nullable ConfigAverageIntegral configOrientation, Class<T> imageType) {
		checkGray(imageType);
		return new DetectDescribeSurfHip<>(BoofHipContext.pack(configDetector), BoofHipContext.pack(configDesc), BoofHipContext.pack(configOrientation), false);
	}

	public static <T extends ImageGray<T>> DetectDescribePoint<T, BrightFeature>
	surfStable(@Nullable ConfigFastHessian configDetector, @Nullable ConfigSurfDescribe.Stability configDescribe, @Nullable ConfigSlidingIntegral configOrientation, Class<T> imageType) {
		checkGray(imageType);
		return new DetectDescribeSurfHip<>(BoofHipContext.pack(configDetector), BoofHipContext.pack(configDescribe), BoofHipContext.pack(configOrientation), true);
	}

	public static DetectDescribePoint<Planar<GrayF32>, BrightFeature>
	surfColorFast(@Nullable ConfigFastHessian configDetector, @Nullable ConfigSurfDescribe.Speed configDesc, @Nullable ConfigAverageIntegral configOrientation,
				  ImageType<Planar<GrayF32>> imageType) {
		if (imageType.getFamily() != ImageType.Family.PLANAR || imageType.getImageClass() != GrayF32.class) throw new IllegalArgumentException("Image type not supported");
		return new DetectDescribeSurfPlanarHip(BoofHipContext.pack(configDetector), BoofHipContext.pack(configDesc), BoofHipContext.pack(configOrientation), false, imageType.getNumBands());
	}

	public static DetectDescribePoint<Planar<GrayF32>, BrightFeature>
	surfColorStable(@Nullable ConfigFastHessian configDetector, @Nullable ConfigSurfDescribe.Stability configDescribe, @Nullable ConfigSlidingIntegral configOrientation,
					ImageType<Planar<GrayF32>> imageType) {
		if (imageType.getFamily() != ImageType.Family.PLANAR || imageType.getImageClass() != GrayF32.class) throw new IllegalArgumentException("Image type not supported");
		return new DetectDescribeSurfPlanarHip(BoofHipContext.pack(configDetector), BoofHipContext.pack(configDescribe), BoofHipContext.pack(configOrientation), true, imageType.getNumBands());
	}

	/** fuseTogether(FactoryInterestPoint.fastHessian(configDetector), null, FactoryDescribeRegionPoint.brief(configBrief, imageType)): the
	 *  definition is made by the reference's own generator on this JVM, exactly as FactoryDescribeRegionPoint.brief does (:187-202). */
	public static <T extends ImageGray<T>> DetectDescribePoint<T, TupleDesc_B>
	fuseTogetherFastHessianBrief(@Nullable ConfigFastHessian configDetector, @Nullable ConfigBrief configBrief, Class<T> imageType) {
		checkGray(imageType);
		if (configBrief == null) configBrief = new ConfigBrief();
		configBrief.checkValidity();
		if (!configBrief.fixed) throw new RuntimeException("the scale / orientation aware BRIEF is not implemented on the GPU");   // caller keeps the Java path
		BinaryCompareDefinition_I32 definition = FactoryBriefDefinition.gaussian2(new Random(123), configBrief.radius, configBrief.numPoints);
		return new DetectDescribeFusionHip<>(BoofHipContext.pack(configDetector), definition);
	}
}
