// jni_replay.cpp -- plays the Scala host (scala/.../VariantsPcaNative.scala) against the REAL JNI shim
// (jni/pcoa_jni.cpp, compiled with tests/jni_stub/jni.h) and libpcoa_hip.so, without a JVM.
//
// Same call sequence as VariantsPcaNative.getSimilarityMatrix / computePca for one GPU task:
//   create -> [per batch of <= `batch` records, in one of two re-used allocPinned slots: CSR buffers -> accumulateCallsEx(CallsPinned
//   | CallsAsync), or carrier bitsets -> accumulateBits for a dense batch; sync every two batches] -> gramFinalize ->
//   commUniqueId -> commInit(rank 0 of 1) -> gramAllreduce -> commDestroy -> gramRead (parity only) -> compute ->
//   timings -> destroy
// Input : <prefix>.idx (int32 LE), <prefix>.offs (int64 LE, n_variants + 1 entries)
// Output: <prefix>.s (int64 N x N), <prefix>.pc (double N x numPc column-major), <prefix>.lam, stdout "nonzero <k>"
// Usage : jni_replay <prefix> <n_samples> <num_pc> [batch = 65536] [mode = 0: the Scala host's rule | 1: lists only | 2: bitsets only]
// Also checks the error mapping the Scala side relies on: an index >= N gives PCOA_ERR_INDEX_RANGE and a message,
// a heap (non-direct) buffer gives PCOA_ERR_INVALID_ARG, num_pc = 0 gives PCOA_ERR_INVALID_ARG.
#include <jni.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>

#include "pcoa.h"

#define FN(name) Java_com_google_cloud_genomics_spark_examples_NativePcoa_00024_##name
extern "C" {
jlong FN(create)(JNIEnv*, jobject, jint, jint, jint);
void FN(destroy)(JNIEnv*, jobject, jlong);
jstring FN(lastError)(JNIEnv*, jobject, jlong);
jint FN(reset)(JNIEnv*, jobject, jlong);
jint FN(accumulateCalls)(JNIEnv*, jobject, jlong, jobject, jobject, jlong);
jint FN(accumulateCallsEx)(JNIEnv*, jobject, jlong, jobject, jobject, jlong, jint);
jint FN(accumulateBits)(JNIEnv*, jobject, jlong, jobject, jlong, jlong);
jint FN(gramFinalize)(JNIEnv*, jobject, jlong);
jint FN(sync)(JNIEnv*, jobject, jlong);
jobject FN(allocPinned)(JNIEnv*, jobject, jlong);
jint FN(freePinned)(JNIEnv*, jobject, jobject);
jint FN(accumulatePlinkBed)(JNIEnv*, jobject, jlong, jobject, jlong, jlong, jint, jint);
jbyteArray FN(commUniqueId)(JNIEnv*, jobject);
jlong FN(commInit)(JNIEnv*, jobject, jlong, jbyteArray, jint, jint);
jint FN(commDestroy)(JNIEnv*, jobject, jlong);
jint FN(commCount)(JNIEnv*, jobject, jlong);
jint FN(gramAllreduce)(JNIEnv*, jobject, jlong, jlong);
jint FN(gramRead)(JNIEnv*, jobject, jlong, jobject);
jint FN(compute)(JNIEnv*, jobject, jlong, jint, jobject, jobject, jintArray);
jint FN(timings)(JNIEnv*, jobject, jlong, jobject);
}

template <typename T>
static std::vector<T> slurp(const std::string& path) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) { std::fprintf(stderr, "cannot read %s\n", path.c_str()); std::exit(2); }
  const std::streamsize bytes = f.tellg();
  f.seekg(0);
  std::vector<T> v((size_t)bytes / sizeof(T));
  f.read(reinterpret_cast<char*>(v.data()), bytes);
  return v;
}
template <typename T>
static void dump(const std::string& path, const std::vector<T>& v) {
  std::ofstream f(path, std::ios::binary);
  f.write(reinterpret_cast<const char*>(v.data()), (std::streamsize)(v.size() * sizeof(T)));
}
#define EXPECT(cond)                                                                  \
  do {                                                                                \
    if (!(cond)) { std::fprintf(stderr, "%s:%d EXPECT(%s) failed\n", __FILE__, __LINE__, #cond); return 1; } \
  } while (0)

int main(int argc, char** argv) {
  if (argc < 4) { std::fprintf(stderr, "usage: jni_replay <prefix> <n_samples> <num_pc> [batch]\n"); return 2; }
  const std::string prefix = argv[1];
  const jint n = std::atoi(argv[2]), num_pc = std::atoi(argv[3]);
  const int64_t batch = argc > 4 ? std::atoll(argv[4]) : 65536;
  const int mode = argc > 5 ? std::atoi(argv[5]) : 0;
  std::vector<int32_t> idx = slurp<int32_t>(prefix + ".idx");
  std::vector<int64_t> offs = slurp<int64_t>(prefix + ".offs");
  const int64_t nv = (int64_t)offs.size() - 1;
  JNIEnv env;
  jobject self = nullptr;  // the NativePcoa$ module instance: unused by the natives

  const jlong ctx = FN(create)(&env, self, n, 0, 0);
  if (ctx == 0) {
    std::fprintf(stderr, "create threw %s: %s\n", env.pending_exception_class.c_str(), env.pending_exception_message.c_str());
    return 3;
  }
  // ---- error mapping first (S stays empty: a rejected call leaves it unchanged)
  {
    int32_t bad_idx[2] = {0, n};
    int64_t bad_offs[2] = {0, 2};
    const jint rc = FN(accumulateCalls)(&env, self, ctx, env.wrapDirect(bad_idx, sizeof(bad_idx)), env.wrapDirect(bad_offs, sizeof(bad_offs)), 1);
    EXPECT(rc == PCOA_ERR_INDEX_RANGE);
    EXPECT(!FN(lastError)(&env, self, ctx)->text.empty());
    EXPECT(FN(accumulateCalls)(&env, self, ctx, env.heapBuffer(), env.heapBuffer(), 1) == PCOA_ERR_INVALID_ARG);
    std::vector<double> c1((size_t)n);
    EXPECT(FN(compute)(&env, self, ctx, 0, env.wrapDirect(c1.data(), 8 * n), nullptr, nullptr) == PCOA_ERR_INVALID_ARG);
    EXPECT(FN(reset)(&env, self, ctx) == PCOA_OK);
  }
  // ---- getSimilarityMatrix, batch by batch exactly as VariantsPcaNative.getSimilarityMatrix does it (r06): two slots of
  // page-locked direct buffers from allocPinned, re-used for the whole partition; a sparse batch goes over as CSR carrier lists
  // through accumulateCallsEx(CallsPinned | CallsAsync) (offsets restart at 0 per batch), a dense one (mean list longer than
  // N / 32 entries, no callset named twice) as carrier bitsets through accumulateBits; one sync per two batches hands both
  // slots back.  mode (argv[5]): 0 = that rule, 1 = every batch as lists, 2 = every batch as bitsets.
  {
    struct Slot {
      jobject idx = nullptr, offs = nullptr, bits = nullptr;
    } slots[2];
    auto fit = [&](jobject& buf, jlong bytes) -> uint8_t* {
      if (buf == nullptr || env.GetDirectBufferCapacity(buf) < bytes) {
        if (buf != nullptr && FN(freePinned)(&env, self, buf) != PCOA_OK) return nullptr;
        buf = FN(allocPinned)(&env, self, std::max<jlong>(bytes + bytes / 4, 1 << 16));
        if (buf == nullptr) return nullptr;
      }
      return static_cast<uint8_t*>(env.GetDirectBufferAddress(buf));
    };
    const int64_t words = ((int64_t)n + 31) / 32;
    int64_t k = 0, as_lists = 0, as_bits = 0;
    for (int64_t v0 = 0; v0 < nv; v0 += batch, ++k) {
      Slot& sl = slots[k & 1];
      if (k >= 2 && (k & 1) == 0) EXPECT(FN(sync)(&env, self, ctx) == PCOA_OK);
      const int64_t rows = std::min(batch, nv - v0);
      const int64_t nnz = offs[(size_t)(v0 + rows)] - offs[(size_t)v0];
      bool bits_ok = mode == 2 || (mode == 0 && nnz > rows * words);
      if (bits_ok) {
        uint32_t* b = reinterpret_cast<uint32_t*>(fit(sl.bits, 4 * words * rows));
        EXPECT(b != nullptr);
        std::memset(b, 0, (size_t)(4 * words * rows));
        for (int64_t r = 0; r < rows; ++r)
          for (int64_t e = offs[(size_t)(v0 + r)]; e < offs[(size_t)(v0 + r + 1)]; ++e) {
            const int32_t c = idx[(size_t)e];
            EXPECT(c >= 0 && c < n);
            uint32_t& w = b[r * words + (c >> 5)];
            if (w & (1u << (c & 31))) bits_ok = false;   // a repeated callset: multiplicities need the lists
            w |= 1u << (c & 31);
          }
        if (bits_ok) {
          EXPECT(FN(accumulateBits)(&env, self, ctx, sl.bits, rows, words) == PCOA_OK);
          as_bits += 1;
        }
      }
      if (!bits_ok) {
        int64_t* bo = reinterpret_cast<int64_t*>(fit(sl.offs, 8 * (rows + 1)));
        int32_t* bi = reinterpret_cast<int32_t*>(fit(sl.idx, 4 * std::max<int64_t>(nnz, 1)));
        EXPECT(bo != nullptr && bi != nullptr);
        for (int64_t r = 0; r <= rows; ++r) bo[r] = offs[(size_t)(v0 + r)] - offs[(size_t)v0];
        if (nnz > 0) std::memcpy(bi, idx.data() + offs[(size_t)v0], (size_t)(4 * nnz));
        EXPECT(FN(accumulateCallsEx)(&env, self, ctx, sl.idx, sl.offs, rows, /*CallsPinned | CallsAsync*/ 2 | 4) == PCOA_OK);
        as_lists += 1;
      }
    }
    EXPECT(FN(gramFinalize)(&env, self, ctx) == PCOA_OK);   // synchronises: the slots are ours again
    for (Slot& sl : slots)
      for (jobject b : {sl.idx, sl.offs, sl.bits})
        if (b != nullptr) EXPECT(FN(freePinned)(&env, self, b) == PCOA_OK);
    std::printf("batches: %lld as lists, %lld as bitsets\n", (long long)as_lists, (long long)as_bits);
  }
  EXPECT(FN(gramFinalize)(&env, self, ctx) == PCOA_OK);
  jbyteArray uid = FN(commUniqueId)(&env, self);
  EXPECT(uid != nullptr && env.GetArrayLength(uid) == 128);
  const jlong comm = FN(commInit)(&env, self, ctx, uid, 0, 1);
  EXPECT(comm != 0);
  EXPECT(FN(commCount)(&env, self, comm) == 1);
  EXPECT(FN(gramAllreduce)(&env, self, ctx, comm) == PCOA_OK);
  EXPECT(FN(commDestroy)(&env, self, comm) == PCOA_OK);
  std::vector<int64_t> s((size_t)n * (size_t)n);
  EXPECT(FN(gramRead)(&env, self, ctx, env.wrapDirect(s.data(), 8 * (jlong)s.size())) == PCOA_OK);
  dump(prefix + ".s", s);
  // ---- computePca
  std::vector<double> comps((size_t)n * (size_t)num_pc), lam((size_t)num_pc);
  jintArray nz = env.newIntArray(1);
  EXPECT(FN(compute)(&env, self, ctx, num_pc, env.wrapDirect(comps.data(), 8 * (jlong)comps.size()),
                     env.wrapDirect(lam.data(), 8 * (jlong)lam.size()), nz) == PCOA_OK);
  dump(prefix + ".pc", comps);
  dump(prefix + ".lam", lam);
  double t3[3] = {0, 0, 0};
  EXPECT(FN(timings)(&env, self, ctx, env.wrapDirect(t3, sizeof(t3))) == PCOA_OK);
  EXPECT((int64_t)t3[0] > 0 || nv == 0);
  std::printf("nonzero %d\n", nz->ints[0]);
  FN(destroy)(&env, self, ctx);
  // ---- the same records as the rows of a PLINK .bed (carrier = heterozygous, the rest homozygous A2 = reference) in
  // page-locked direct buffers from allocPinned, queued in three pieces (BedHostAsync = 2): the S of the CSR path again
  {
    const jlong bpv = ((jlong)n + 3) / 4;
    const int64_t cut[4] = {0, nv / 3, nv / 3 + (nv - nv / 3) / 2, nv};
    const jlong c2 = FN(create)(&env, self, n, 0, 0);
    EXPECT(c2 != 0);
    jobject pin[3] = {nullptr, nullptr, nullptr};
    for (int q = 0; q < 3; ++q) {
      const int64_t rows = cut[q + 1] - cut[q];
      if (rows == 0) continue;
      pin[q] = FN(allocPinned)(&env, self, rows * bpv);
      EXPECT(pin[q] != nullptr && env.GetDirectBufferCapacity(pin[q]) == rows * bpv);
      uint8_t* b = static_cast<uint8_t*>(env.GetDirectBufferAddress(pin[q]));
      std::memset(b, 0xFF, (size_t)(rows * bpv));                       // every sample homozygous A2
      for (int64_t r = 0; r < rows; ++r)
        for (int64_t e = offs[(size_t)(cut[q] + r)]; e < offs[(size_t)(cut[q] + r + 1)]; ++e) {
          const int32_t c = idx[(size_t)e];
          b[r * bpv + c / 4] = (uint8_t)(b[r * bpv + c / 4] & ~(1u << (2 * (c % 4))));   // 11 -> 10: heterozygous
        }
      EXPECT(FN(accumulatePlinkBed)(&env, self, c2, pin[q], rows, bpv, 0, 2) == PCOA_OK);
    }
    EXPECT(FN(accumulatePlinkBed)(&env, self, c2, env.heapBuffer(), 1, bpv, 0, 0) == PCOA_ERR_INVALID_ARG);
    EXPECT(FN(sync)(&env, self, c2) == PCOA_OK);                           // the queued rows are the caller's again
    for (int q = 0; q < 3; ++q)
      if (pin[q]) EXPECT(FN(freePinned)(&env, self, pin[q]) == PCOA_OK);
    EXPECT(FN(gramFinalize)(&env, self, c2) == PCOA_OK);
    std::vector<int64_t> s2((size_t)n * (size_t)n);
    EXPECT(FN(gramRead)(&env, self, c2, env.wrapDirect(s2.data(), 8 * (jlong)s2.size())) == PCOA_OK);
    EXPECT(s2 == s);
    FN(destroy)(&env, self, c2);
    std::printf("plink rows through the shim: same S\n");
  }
  return 0;
}
