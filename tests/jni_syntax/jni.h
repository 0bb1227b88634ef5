/* Declarations-only stand-in for the JDK's <jni.h>, used ONLY by tests/test_jni_shim.py to syntax-check the generated
 * integration/jni/boofhip_jni.c with `gcc -fsyntax-only` in an image that has no JDK.  It declares the handful of JNI types and the
 * JNIEnv members the shim uses, with the signatures of the JNI specification; nothing here is ever linked or run.  A real build uses
 * $JAVA_HOME/include/jni.h. */
#ifndef BHIP_TEST_JNI_H
#define BHIP_TEST_JNI_H
#include <stdint.h>
typedef int32_t jint; typedef int64_t jlong; typedef int8_t jbyte; typedef int16_t jshort; typedef float jfloat; typedef double jdouble;
typedef uint8_t jboolean; typedef jint jsize;
typedef struct _jobject* jobject;
typedef jobject jclass; typedef jobject jstring; typedef jobject jarray; typedef jarray jobjectArray; typedef jarray jintArray; typedef jarray jlongArray;
typedef jarray jbyteArray; typedef jarray jshortArray; typedef jarray jfloatArray; typedef jarray jdoubleArray;
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;
struct JNINativeInterface_ {
	jsize (*GetArrayLength)(JNIEnv*, jarray);
	jobject (*GetObjectArrayElement)(JNIEnv*, jobjectArray, jsize);
	void* (*GetPrimitiveArrayCritical)(JNIEnv*, jarray, jboolean*);
	void (*ReleasePrimitiveArrayCritical)(JNIEnv*, jarray, void*, jint);
	void (*SetLongArrayRegion)(JNIEnv*, jlongArray, jsize, jsize, const jlong*);
	void* (*GetDirectBufferAddress)(JNIEnv*, jobject);
	jstring (*NewStringUTF)(JNIEnv*, const char*);
	jobject (*NewDirectByteBuffer)(JNIEnv*, void*, jlong);
};
#endif
