package boofcv.hip;

import boofcv.abst.feature.associate.AssociateDescription;
import boofcv.abst.feature.associate.ScoreAssociateEuclideanSq_F64;
import boofcv.abst.feature.associate.ScoreAssociateEuclidean_F64;
import boofcv.abst.feature.associate.ScoreAssociateHamming_B;
import boofcv.abst.feature.associate.ScoreAssociation;

/** FactoryAssociation.greedy(score, maxErrorThreshold, backwardsValidation) (main/boofcv-feature/.../factory/feature/associate/FactoryAssociation.java:51-65)
 *  for the three scores the GPU implements; anything else is declined so the caller keeps the Java AssociateGreedy. */
public class FactoryAssociationHip {
	@SuppressWarnings("unchecked")
	public static <D> AssociateDescription<D> greedy(ScoreAssociation<D> score, double maxErrorThreshold, boolean backwardsValidation) {
		int kind;
		if (score instanceof ScoreAssociateEuclideanSq_F64) kind = AssociateGreedyHip.L2_SQ;
		else if (score instanceof ScoreAssociateEuclidean_F64) kind = AssociateGreedyHip.L2;
		else if (score instanceof ScoreAssociateHamming_B) kind = AssociateGreedyHip.HAMMING;
		else throw new RuntimeException("score type not implemented on the GPU");
		return (AssociateDescription<D>)new AssociateGreedyHip(kind, score, maxErrorThreshold, backwardsValidation);
	}
}
