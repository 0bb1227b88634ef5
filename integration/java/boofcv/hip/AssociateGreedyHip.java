package boofcv.hip;

import org.ddogleg.struct.FastQueue;
import org.ddogleg.struct.GrowQueue_I32;

import boofcv.abst.feature.associate.AssociateDescription;
import boofcv.abst.feature.associate.ScoreAssociation;
import boofcv.struct.feature.AssociatedIndex;
import boofcv.struct.feature.MatchScoreType;
import boofcv.struct.feature.TupleDesc_B;
import boofcv.struct.feature.TupleDesc_F64;

/** AssociateDescription over bhip_assoc_l2_f64 / bhip_assoc_hamming: WrapAssociateGreedy.java:73-123 + AssociateGreedy.java:65-118 + FindUnassociated.java:38-72.
 *  Descriptor lists are packed into one contiguous array per call (the reference stores one Java array per feature). */
@SuppressWarnings("rawtypes")
public class AssociateGreedyHip implements AssociateDescription {
	static final int L2_SQ = 0, L2 = 1, HAMMING = 2;
	private final long ctx = BoofHipContext.get();
	private final int kind;
	private final ScoreAssociation score;
	private double maxErr;
	private final boolean backwards;
	private FastQueue src, dst;
	private int[] pairs = new int[0];
	private double[] fit = new double[0];
	private final FastQueue<AssociatedIndex> matches = new FastQueue<>(10, AssociatedIndex.class, true);
	private final GrowQueue_I32 unassocSrc = new GrowQueue_I32(), unassocDst = new GrowQueue_I32();

	AssociateGreedyHip(int kind, ScoreAssociation score, double maxErr, boolean backwards) { this.kind = kind; this.score = score; this.maxErr = maxErr; this.backwards = backwards; }

	@Override public void setSource(FastQueue listSrc) { this.src = listSrc; }
	@Override public void setDestination(FastQueue listDst) { this.dst = listDst; }

	private static double[] packF64(FastQueue q, int dof) {
		double[] a = new double[q.size*dof];
		for (int i = 0; i < q.size; i++) System.arraycopy(((TupleDesc_F64)q.get(i)).value, 0, a, i*dof, dof);
		return a;
	}
	private static int[] packB(FastQueue q, int words) {
		int[] a = new int[q.size*words];
		for (int i = 0; i < q.size; i++) System.arraycopy(((TupleDesc_B)q.get(i)).data, 0, a, i*words, words);
		return a;
	}

	@Override public void associate() {
		final int ns = src.size, nd = dst.size;
		if (pairs.length < ns) { pairs = new int[ns]; fit = new double[ns]; }
		if (ns > 0) {
			if (kind == HAMMING) {
				int words = ((TupleDesc_B)src.get(0)).data.length;
				BoofHip.check(ctx, BoofHip.assocHamming(ctx, packB(src, words), ns, nd > 0 ? packB(dst, words) : null, nd, words, maxErr, backwards ? 1 : 0, pairs, fit));
			} else {
				int dof = ((TupleDesc_F64)src.get(0)).value.length;
				BoofHip.check(ctx, BoofHip.assocL2F64(ctx, packF64(src, dof), ns, nd > 0 ? packF64(dst, dof) : null, nd, dof, maxErr, backwards ? 1 : 0, kind == L2 ? 1 : 0, pairs, fit));
			}
		}
		// WrapAssociateGreedy.associate: matches in increasing source index, then the unassociated lists
		matches.reset(); unassocSrc.reset(); unassocDst.reset();
		boolean[] matched = new boolean[nd];
		for (int i = 0; i < ns; i++) {
			if (pairs[i] >= 0) { matches.grow().setAssociation(i, pairs[i], fit[i]); matched[pairs[i]] = true; }
			else unassocSrc.add(i);
		}
		for (int j = 0; j < nd; j++) if (!matched[j]) unassocDst.add(j);
	}

	@Override public FastQueue<AssociatedIndex> getMatches() { return matches; }
	@Override public GrowQueue_I32 getUnassociatedSource() { return unassocSrc; }
	@Override public GrowQueue_I32 getUnassociatedDestination() { return unassocDst; }
	@Override public void setMaxScoreThreshold(double score) { this.maxErr = score; }
	@Override public MatchScoreType getScoreType() { return score.getScoreType(); }
	@Override public boolean uniqueSource() { return true; }
	@Override public boolean uniqueDestination() { return backwards; }
}
