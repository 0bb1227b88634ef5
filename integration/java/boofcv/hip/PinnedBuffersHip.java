package boofcv.hip;

import java.nio.ByteBuffer;
import java.nio.ByteOrder;
import java.nio.DoubleBuffer;
import java.util.ArrayDeque;
import java.util.HashMap;
import java.util.Map;

/**
 * Page-locked host memory (bhip_host_alloc / bhip_host_free) as direct buffers, pooled by size class: the store a provider keeps its fetched
 * descriptor lists and result arrays in, so that bhip_surf_fetch and the descriptor uploads of bhip_assoc_l2_f64 are DMA transfers (the Python
 * mirror does the same: boofcv_amd/api.py, _PinnedPool).  Falls back to ordinary direct buffers when no page-locked memory can be had.
 * Natives: integration/jni/boofhip_jni_buffers.c.  Not thread safe: one pool per provider object, like its context.
 */
public final class PinnedBuffersHip implements AutoCloseable {
	static { System.loadLibrary("boofhip_jni"); }

	private static native ByteBuffer allocate(long ctx, long bytes);
	private static native int release(ByteBuffer buffer);
	static native long address(ByteBuffer buffer);

	private final long ctx;
	private final Map<Integer, ArrayDeque<ByteBuffer>> free = new HashMap<>();
	private long pooledBytes;
	private static final long MAX_POOLED = 1L << 30;

	public PinnedBuffersHip(long ctx) { this.ctx = ctx; }

	/** powers of two and the sizes half way between them, from 4 KiB */
	static int sizeClass(long bytes) {
		long c = 4096;
		while (c < bytes) c = (c & (c - 1)) == 0 ? c * 3 / 2 : c * 4 / 3;
		if (c > Integer.MAX_VALUE) throw new IllegalArgumentException("buffer too large");
		return (int)c;
	}

	/** a buffer of at least `bytes` bytes, native byte order, position 0 */
	public ByteBuffer take(long bytes) {
		final int size = sizeClass(Math.max(bytes, 1));
		ArrayDeque<ByteBuffer> q = free.get(size);
		ByteBuffer b = q != null ? q.pollLast() : null;
		if (b != null) pooledBytes -= size;
		else {
			b = allocate(ctx, size);
			if (b == null) b = ByteBuffer.allocateDirect(size);   // pageable, still a valid host pointer for the C ABI
		}
		b.clear();
		return b.order(ByteOrder.nativeOrder());
	}

	/** descriptor list of n features of `dof` doubles */
	public DoubleBuffer takeDoubles(int n, int dof) { return take(8L * n * dof).asDoubleBuffer(); }

	/** hands a buffer made by take() back to the pool */
	public void give(ByteBuffer b) {
		final int size = b.capacity();
		if (pooledBytes + size > MAX_POOLED) { release(b); return; }
		free.computeIfAbsent(size, k -> new ArrayDeque<>()).addLast(b);
		pooledBytes += size;
	}

	@Override public void close() {
		for (ArrayDeque<ByteBuffer> q : free.values()) for (ByteBuffer b : q) release(b);
		free.clear();
		pooledBytes = 0;
	}
}
