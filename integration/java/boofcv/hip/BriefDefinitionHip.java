package boofcv.hip;

import boofcv.alg.feature.describe.brief.BinaryCompareDefinition_I32;

/** Flattens a BinaryCompareDefinition_I32 (main/boofcv-feature/.../alg/feature/describe/brief/BinaryCompareDefinition_I32.java:37-56) into the
 *  int arrays the C ABI takes: samplePoints[numSamples][2], compare[numPairs][2].  The definition itself is made by the reference's own
 *  FactoryBriefDefinition.gaussian2(new Random(123), radius, numPoints) on this JVM (java.util.Random + StrictMath), so the table is the
 *  reference's bit for bit.  UNCOMPILED SOURCE. */
final class BriefDefinitionHip {
	final int radius, numPoints;
	final int[] samplePoints, compare;

	BriefDefinitionHip(BinaryCompareDefinition_I32 def) {
		radius = def.radius;
		numPoints = def.compare.length;
		samplePoints = new int[2*def.samplePoints.length];
		compare = new int[2*def.compare.length];
		for (int i = 0; i < def.samplePoints.length; i++) { samplePoints[2*i] = def.samplePoints[i].x; samplePoints[2*i + 1] = def.samplePoints[i].y; }
		for (int i = 0; i < def.compare.length; i++) { compare[2*i] = def.compare[i].x; compare[2*i + 1] = def.compare[i].y; }
	}

	int words() { return (numPoints + 31)/32; }
}
