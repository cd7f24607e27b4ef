// pcoa_jni.cpp -- JNI shim between the Scala host (scala/.../NativePcoa.scala) and the C ABI of include/pcoa.h.
//
// SURVEY.md 8(b) caller (3) / 8(f) rank 4; north_star: "a Scala host calling hand-written HIP kernels through a thin
// JNI/C-ABI layer".  Every native is a forward to ONE pcoa_* call: buffers are direct java.nio.ByteBuffers
// (little-endian, allocated by NativePcoa.direct) whose addresses go to the C ABI as they are -- no copies, no JVM
// arrays on the hot path.  A ctx / communicator handle travels as a jlong.  Errors: the natives return the pcoa_status
// (0 = ok) and the Scala side maps it to the exception the reference would have thrown ("Error
// mapping"); only create() throws by itself, because it has no handle to return a message through.
//
// Replaces, together with NativePcoa.scala / VariantsPcaNative.scala:
//   VariantsPcaDriver.getSimilarityMatrix  (VariantsPca.scala:182-191)  create, accumulateCalls | accumulateBits,
//                                                                        gramFinalize, comm*, gramAllreduce
//   VariantsPcaDriver.computePca           (VariantsPca.scala:198-231)  compute
//   VariantsPcaDriver.stop                 (VariantsPca.scala:283-285)  destroy
//
// Build where a JDK exists:  make -C jni JAVA_HOME=/path/to/jdk      (-> jni/libpcoa_jni.so)
// Without a JDK the same file is compiled and RUN against tests/jni_stub/jni.h (a minimal JNIEnv over plain memory):
// tests/jni_replay.cpp replays the call sequence of VariantsPcaNative.scala through these natives.
#include <jni.h>

#include <cstdint>
#include <cstring>

#include "pcoa.h"

#define FN(name) Java_com_google_cloud_genomics_spark_examples_NativePcoa_00024_##name

namespace {
inline pcoa_ctx* ctx_of(jlong h) { return reinterpret_cast<pcoa_ctx*>(static_cast<intptr_t>(h)); }

// address of a direct buffer, or nullptr (and PCOA_ERR_INVALID_ARG from the C ABI) for a heap buffer
template <typename T>
inline T* direct(JNIEnv* env, jobject buf) {
  return buf ? static_cast<T*>(env->GetDirectBufferAddress(buf)) : nullptr;
}
}  // namespace

extern "C" {

// pcoa_create.  Throws IllegalStateException with pcoa_last_error(NULL) when no engine can be made (no HIP device:
// there is no CPU fallback).
JNIEXPORT jlong JNICALL FN(create)(JNIEnv* env, jobject, jint n_samples, jint device, jint flags) {
  pcoa_ctx* c = nullptr;
  if (pcoa_create(&c, n_samples, device, static_cast<uint32_t>(flags)) != PCOA_OK) {
    jclass ex = env->FindClass("java/lang/IllegalStateException");
    if (ex) env->ThrowNew(ex, pcoa_last_error(nullptr));
    return 0;
  }
  return static_cast<jlong>(reinterpret_cast<intptr_t>(c));
}

// pcoa_destroy
JNIEXPORT void JNICALL FN(destroy)(JNIEnv*, jobject, jlong ctx) { pcoa_destroy(ctx_of(ctx)); }

// pcoa_last_error
JNIEXPORT jstring JNICALL FN(lastError)(JNIEnv* env, jobject, jlong ctx) {
  const char* m = pcoa_last_error(ctx_of(ctx));
  return env->NewStringUTF(m ? m : "");
}

// pcoa_reset
JNIEXPORT jint JNICALL FN(reset)(JNIEnv*, jobject, jlong ctx) { return pcoa_reset(ctx_of(ctx)); }

// pcoa_accumulate_calls: one batch of RDD[Seq[Int]] records as CSR (sampleIdx: int32[nnz], rowOffsets: int64[n + 1])
JNIEXPORT jint JNICALL FN(accumulateCalls)(JNIEnv* env, jobject, jlong ctx, jobject sample_idx, jobject row_offsets,
                                           jlong n_variants) {
  return pcoa_accumulate_calls(ctx_of(ctx), direct<const int32_t>(env, sample_idx), direct<const int64_t>(env, row_offsets),
                               static_cast<int64_t>(n_variants));
}

// pcoa_accumulate_calls_ex: the same batch with the caller saying where the buffers live and whether it will wait
// (PCOA_CALLS_HOST_PINNED | PCOA_CALLS_ASYNC for a host that builds batch k+1 while batch k travels)
JNIEXPORT jint JNICALL FN(accumulateCallsEx)(JNIEnv* env, jobject, jlong ctx, jobject sample_idx, jobject row_offsets,
                                             jlong n_variants, jint flags) {
  return pcoa_accumulate_calls_ex(ctx_of(ctx), direct<const int32_t>(env, sample_idx), direct<const int64_t>(env, row_offsets),
                                  static_cast<int64_t>(n_variants), static_cast<uint32_t>(flags));
}

// pcoa_sync: the synchronising call after which asynchronously handed-over buffers may be reused
JNIEXPORT jint JNICALL FN(sync)(JNIEnv*, jobject, jlong ctx) { return pcoa_sync(ctx_of(ctx)); }

// pcoa_gram_reduce_from: dst.S += src.S, two engines of this JVM
JNIEXPORT jint JNICALL FN(gramReduceFrom)(JNIEnv*, jobject, jlong dst, jlong src) {
  return pcoa_gram_reduce_from(ctx_of(dst), ctx_of(src));
}

// pcoa_accumulate_bits: one carrier bitset per record (java.util.BitSet.toLongArray() written little-endian)
JNIEXPORT jint JNICALL FN(accumulateBits)(JNIEnv* env, jobject, jlong ctx, jobject bits, jlong n_variants, jlong ld_words) {
  return pcoa_accumulate_bits(ctx_of(ctx), direct<const uint32_t>(env, bits), static_cast<int64_t>(n_variants),
                              static_cast<int64_t>(ld_words), /*is_device_ptr=*/0);
}

// pcoa_host_alloc_pinned as a direct ByteBuffer over page-locked memory (null when it cannot be had): what
// accumulateCallsEx(.. CallsPinned ..) and accumulatePlinkBed(.. BedHostAsync) read at link speed.  freePinned releases it;
// the buffer must not be used afterwards (the JVM does not own this memory and never frees it by itself).
JNIEXPORT jobject JNICALL FN(allocPinned)(JNIEnv* env, jobject, jlong bytes) {
  void* p = nullptr;
  if (bytes <= 0 || pcoa_host_alloc_pinned(static_cast<size_t>(bytes), &p) != PCOA_OK) return nullptr;
  jobject buf = env->NewDirectByteBuffer(p, bytes);
  if (!buf) (void)pcoa_host_free_pinned(p);
  return buf;
}
JNIEXPORT jint JNICALL FN(freePinned)(JNIEnv* env, jobject, jobject buf) {
  return pcoa_host_free_pinned(direct<void>(env, buf));
}

// pcoa_accumulate_plink_bed: the rows of a variant-major PLINK .bed as they lie in the file; mode 0 host rows (consumed on
// return), 1 device address, 2 = PCOA_BED_HOST_ASYNC (page-locked rows, only queued)
JNIEXPORT jint JNICALL FN(accumulatePlinkBed)(JNIEnv* env, jobject, jlong ctx, jobject rows, jlong n_variants, jlong row_bytes,
                                              jint ref_is_a1, jint mode) {
  return pcoa_accumulate_plink_bed(ctx_of(ctx), direct<const uint8_t>(env, rows), n_variants, row_bytes, ref_is_a1, mode);
}

// pcoa_gram_finalize
JNIEXPORT jint JNICALL FN(gramFinalize)(JNIEnv*, jobject, jlong ctx) { return pcoa_gram_finalize(ctx_of(ctx)); }

// pcoa_comm_unique_id: 128 bytes for rank 0 to broadcast (sc.broadcast), or null on failure
JNIEXPORT jbyteArray JNICALL FN(commUniqueId)(JNIEnv* env, jobject) {
  uint8_t id[128];
  if (pcoa_comm_unique_id(id) != PCOA_OK) return nullptr;
  jbyteArray out = env->NewByteArray(128);
  if (out) env->SetByteArrayRegion(out, 0, 128, reinterpret_cast<const jbyte*>(id));
  return out;
}

// pcoa_comm_init: returns the communicator handle, 0 on failure (message in lastError(ctx))
JNIEXPORT jlong JNICALL FN(commInit)(JNIEnv* env, jobject, jlong ctx, jbyteArray id, jint rank, jint n_ranks) {
  if (!id || env->GetArrayLength(id) != 128) return 0;
  uint8_t raw[128];
  env->GetByteArrayRegion(id, 0, 128, reinterpret_cast<jbyte*>(raw));
  void* comm = nullptr;
  if (pcoa_comm_init(ctx_of(ctx), raw, rank, n_ranks, &comm) != PCOA_OK) return 0;
  return static_cast<jlong>(reinterpret_cast<intptr_t>(comm));
}

// pcoa_comm_destroy
JNIEXPORT jint JNICALL FN(commDestroy)(JNIEnv*, jobject, jlong comm) {
  return pcoa_comm_destroy(reinterpret_cast<void*>(static_cast<intptr_t>(comm)));
}

// pcoa_comm_count: ncclCommCount of the communicator, or a negative pcoa_status (what the host logs to show the collective
// spans the GPUs it was started on)
JNIEXPORT jint JNICALL FN(commCount)(JNIEnv*, jobject, jlong comm) {
  int32_t n = 0;
  const int rc = pcoa_comm_count(reinterpret_cast<void*>(static_cast<intptr_t>(comm)), &n);
  return rc == PCOA_OK ? static_cast<jint>(n) : static_cast<jint>(rc);
}

// pcoa_gram_allreduce_rccl: reduceByKey(_ + _) over the GPUs (VariantsPca.scala:190)
JNIEXPORT jint JNICALL FN(gramAllreduce)(JNIEnv*, jobject, jlong ctx, jlong comm) {
  return pcoa_gram_allreduce_rccl(ctx_of(ctx), reinterpret_cast<void*>(static_cast<intptr_t>(comm)));
}

// pcoa_gram_read_i64: all N^2 entries as int64 (checkpoints; the ((Int, Int), Int) entries of :189 if a caller wants them)
JNIEXPORT jint JNICALL FN(gramRead)(JNIEnv* env, jobject, jlong ctx, jobject out_nxn) {
  return pcoa_gram_read_i64(ctx_of(ctx), direct<int64_t>(env, out_nxn));
}

// pcoa_gram_load_i64: resume from a checkpoint / enter at computePca(matrixEntries)
JNIEXPORT jint JNICALL FN(gramLoad)(JNIEnv* env, jobject, jlong ctx, jobject in_nxn) {
  return pcoa_gram_load_i64(ctx_of(ctx), direct<const int64_t>(env, in_nxn));
}

// pcoa_compute: components = N x numPc doubles column-major (== pca.toArray, :227), eigenvalues = numPc doubles
// (may be null), nonZeroRows = int[1] (may be null)
JNIEXPORT jint JNICALL FN(compute)(JNIEnv* env, jobject, jlong ctx, jint num_pc, jobject components, jobject eigenvalues,
                                   jintArray non_zero_rows) {
  int32_t nonzero = 0;
  const int rc = pcoa_compute(ctx_of(ctx), num_pc, direct<double>(env, components), direct<double>(env, eigenvalues),
                              &nonzero);
  if (non_zero_rows && env->GetArrayLength(non_zero_rows) >= 1) {
    const jint v = nonzero;
    env->SetIntArrayRegion(non_zero_rows, 0, 1, &v);
  }
  return rc;
}

// pcoa_get_timings: what reportIoStats prints (:48, :281) -- {variants, Gram kernel seconds, PCoA seconds}
JNIEXPORT jint JNICALL FN(timings)(JNIEnv* env, jobject, jlong ctx, jobject out3_doubles) {
  pcoa_timings t;
  const int rc = pcoa_get_timings(ctx_of(ctx), &t);
  double* out = direct<double>(env, out3_doubles);
  if (rc == PCOA_OK && out) {
    out[0] = static_cast<double>(t.gram_variants);
    out[1] = t.gram_kernel_seconds + t.pack_seconds;
    out[2] = t.compute_total_seconds;
  }
  return rc;
}

}  // extern "C"
