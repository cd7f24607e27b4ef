/* Minimal stand-in for <jni.h> (test infrastructure, NOT a JDK header): just enough of the JNI C++ surface for
 * jni/pcoa_jni.cpp to compile and RUN without a JVM.  Objects are small tagged records over plain memory; a "direct
 * ByteBuffer" is {address, capacity}.  tests/jni_replay.cpp plays the Scala host against it. */
#ifndef PCOA_TEST_JNI_STUB_H_
#define PCOA_TEST_JNI_STUB_H_

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL

typedef int32_t jint;
typedef int64_t jlong;
typedef int8_t jbyte;
typedef int32_t jsize;

struct _jobject {
  enum Kind { DIRECT_BUFFER, HEAP_BUFFER, BYTE_ARRAY, INT_ARRAY, STRING, CLASS } kind;
  void* address = nullptr;        // DIRECT_BUFFER
  jlong capacity = 0;
  std::vector<jbyte> bytes;       // BYTE_ARRAY
  std::vector<jint> ints;         // INT_ARRAY
  std::string text;               // STRING / CLASS name
};
typedef _jobject* jobject;
typedef jobject jclass;
typedef jobject jstring;
typedef jobject jbyteArray;
typedef jobject jintArray;

struct JNIEnv {
  std::vector<_jobject*> owned;   // freed by the test at exit
  std::string pending_exception_class, pending_exception_message;

  ~JNIEnv() { for (auto* o : owned) delete o; }
  jobject make(_jobject::Kind k) { auto* o = new _jobject(); o->kind = k; owned.push_back(o); return o; }

  void* GetDirectBufferAddress(jobject buf) { return (buf && buf->kind == _jobject::DIRECT_BUFFER) ? buf->address : nullptr; }
  jlong GetDirectBufferCapacity(jobject buf) { return (buf && buf->kind == _jobject::DIRECT_BUFFER) ? buf->capacity : -1; }
  jobject NewDirectByteBuffer(void* p, jlong cap) { return p ? wrapDirect(p, cap) : nullptr; }
  jbyteArray NewByteArray(jsize n) { jobject o = make(_jobject::BYTE_ARRAY); o->bytes.assign((size_t)n, 0); return o; }
  jsize GetArrayLength(jobject a) { return a->kind == _jobject::BYTE_ARRAY ? (jsize)a->bytes.size() : (jsize)a->ints.size(); }
  void SetByteArrayRegion(jbyteArray a, jsize start, jsize len, const jbyte* src) { std::memcpy(a->bytes.data() + start, src, (size_t)len); }
  void GetByteArrayRegion(jbyteArray a, jsize start, jsize len, jbyte* dst) { std::memcpy(dst, a->bytes.data() + start, (size_t)len); }
  void SetIntArrayRegion(jintArray a, jsize start, jsize len, const jint* src) { std::memcpy(a->ints.data() + start, src, sizeof(jint) * (size_t)len); }
  jstring NewStringUTF(const char* s) { jobject o = make(_jobject::STRING); o->text = s ? s : ""; return o; }
  jclass FindClass(const char* name) { jobject o = make(_jobject::CLASS); o->text = name; return o; }
  jint ThrowNew(jclass cls, const char* msg) { pending_exception_class = cls->text; pending_exception_message = msg ? msg : ""; return 0; }

  // helpers of the test side (what java.nio / scala would do)
  jobject wrapDirect(void* p, jlong cap) { jobject o = make(_jobject::DIRECT_BUFFER); o->address = p; o->capacity = cap; return o; }
  jobject heapBuffer() { return make(_jobject::HEAP_BUFFER); }
  jintArray newIntArray(jsize n) { jobject o = make(_jobject::INT_ARRAY); o->ints.assign((size_t)n, 0); return o; }
};

#endif
