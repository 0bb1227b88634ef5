/* Hand-written companion of the generated shim: page-locked host memory (bhip_host_alloc) as java.nio direct buffers, so that frames,
 * fetched results and descriptor lists cross the boundary by DMA instead of through GetPrimitiveArrayCritical + a staged pageable copy
 * ("Page-locked memory for what crosses the boundary").  Uncompiled here (no JDK in the image); syntax-checked by
 * tests/test_jni_shim.py against the JNI signatures. */
#include <jni.h>
#include <stddef.h>
#include <stdint.h>
#include "boofhip.h"

#define BHIP_JNI(ret, name) JNIEXPORT ret JNICALL Java_boofcv_hip_PinnedBuffersHip_##name

/* ByteBuffer over `bytes` bytes of page-locked memory (null when the allocation fails: the caller falls back to ordinary arrays) */
BHIP_JNI(jobject, allocate)(JNIEnv* e, jclass cls, jlong ctx, jlong bytes) {
	uint8_t* p = NULL;
	if (bhip_host_alloc((bhip_ctx*)(intptr_t)ctx, (long long)bytes, &p) != 0 || !p) return NULL;
	jobject buf = (*e)->NewDirectByteBuffer(e, p, bytes);
	if (!buf) (void)bhip_host_free(p);
	return buf;
}

/* releases the block behind a buffer made by allocate(); the buffer must not be used afterwards */
BHIP_JNI(jint, release)(JNIEnv* e, jclass cls, jobject buffer) {
	void* p = buffer ? (*e)->GetDirectBufferAddress(e, buffer) : NULL;
	return (jint)bhip_host_free(p);
}

/* the address a `dev_`-style long argument of BoofHip wants when a direct buffer is handed to an entry point that takes a host pointer
 * through a GetDirectBufferAddress variant of the shim (the C ABI takes plain pointers either way) */
BHIP_JNI(jlong, address)(JNIEnv* e, jclass cls, jobject buffer) {
	return (jlong)(intptr_t)(buffer ? (*e)->GetDirectBufferAddress(e, buffer) : NULL);
}
