// exp_bits.hip -- correctness + A/B harness of the k-bits operand (spark-examples_amd/csrc/gram_kbits.inl).
//
// On one MI355X, with the library's own kernels (gram_packed.hip is included, so the anonymous-namespace kernels are visible):
//   1. small ragged shapes: every k-bits pre-pass (fp32 / uint8 / carrier bitsets / CSR) against a host-built operand, and the
//      k-bits contraction (split-K, lock-step, even-split launches) against a host popcount Gram;
//   2. BASELINE configs[1] size (2,504 samples x 2^20 variants): S of the k-bits path == S of the shipped FP4 path, bit for bit;
//   3. times per launch: pre-passes (FP4 vs k-bits, per input format), contractions (FP4 lock-step / split-K vs k-bits lock-step /
//      even split / split-K, whole chip and half chip), and the pipelined step (pre-pass beside the contraction of the other buffer).
// Not part of the product.  Build: make -C tools exp_bits.  Prints one line per measurement.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../spark-examples_amd/csrc/gram_packed.hip"

namespace pcoa {
const DebugKnobs& debug_knobs() {
  static const DebugKnobs k;
  return k;
}
}  // namespace pcoa

using namespace pcoa;

#define CK(expr)                                                                                  \
  do {                                                                                            \
    hipError_t e_ = (expr);                                                                       \
    if (e_ != hipSuccess) {                                                                       \
      std::fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
      std::exit(1);                                                                               \
    }                                                                                             \
  } while (0)

__global__ void fill_kernel(float* x, int64_t rows, int64_t n, int64_t ld, uint32_t seed, uint32_t thr) {
  const int64_t count = rows * ld;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t col = i % ld;
    uint32_t h = (uint32_t)i * 2654435761u ^ (uint32_t)(i >> 32) * 40503u ^ seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    x[i] = col < n ? ((h < thr) ? 1.0f : 0.0f) : 7.5f;  // padding columns hold garbage on purpose
  }
}
__global__ void to_u8_kernel(const float* x, uint8_t* y, int64_t rows, int64_t n, int64_t ld, int64_t ld8) {
  const int64_t count = rows * ld8;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / ld8, c = i % ld8;
    y[i] = c < n ? (uint8_t)x[r * ld + c] : (uint8_t)0xff;
  }
}
// natural carrier bitsets: row v, bit (i & 31) of word i >> 5 = sample i; words beyond the row hold garbage
__global__ void to_bits_kernel(const float* x, uint32_t* bits, int64_t rows, int64_t n, int64_t ld, int64_t ldw) {
  const int64_t count = rows * ldw;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / ldw, w = i % ldw;
    uint32_t v = 0;
    for (int b = 0; b < 32; ++b) {
      const int64_t c = w * 32 + b;
      if (c < n) v |= (x[r * ld + c] == 1.0f ? 1u : 0u) << b;
      else v |= (uint32_t)((c * 7 + r) & 1) << b;  // garbage beyond N
    }
    bits[i] = v;
  }
}
__global__ void diff_kernel(const uint32_t* a, const uint32_t* b, int64_t words, unsigned long long* out) {
  unsigned long long d = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (int64_t)gridDim.x * blockDim.x)
    d += (a[i] != b[i]);
  if (d) atomicAdd(out, d);
}
__global__ void delay_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static unsigned long long count_diff(const void* a, const void* b, int64_t bytes, unsigned long long* cnt) {
  CK(hipMemset(cnt, 0, 8));
  hipLaunchKernelGGL(diff_kernel, dim3(2048), dim3(256), 0, 0, (const uint32_t*)a, (const uint32_t*)b, bytes / 4, cnt);
  unsigned long long h = 0;
  CK(hipMemcpy(&h, cnt, 8, hipMemcpyDeviceToHost));
  return h;
}

static float time_ms(hipStream_t s, int reps, const std::function<void()>& body) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  body();  // warm-up
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(a, s));
  for (int r = 0; r < reps; ++r) body();
  CK(hipEventRecord(b, s));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  CK(hipEventDestroy(a));
  CK(hipEventDestroy(b));
  return ms / reps;
}

// ---- part 1: small ragged shapes against the host ------------------------------------------------------------------------
static int small_case(int n, int64_t v, int64_t ld, uint32_t thr, int num_cu, unsigned long long* cnt) {
  const int npad = (int)gram_packed_npad(n);
  const int64_t nblk = gram_kb_pad(v, 2) / 4;
  const int64_t kbytes = nblk * npad * 16;
  float* x;
  CK(hipMalloc(&x, (size_t)(v * ld * 4 + 64)));
  hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, x, v, (int64_t)n, ld, 12345u + (uint32_t)n, thr);
  std::vector<float> hx((size_t)(v * ld));
  CK(hipMemcpy(hx.data(), x, (size_t)(v * ld * 4), hipMemcpyDeviceToHost));
  // host operand + host Gram (column bitsets, popcount)
  std::vector<uint32_t> hk((size_t)(kbytes / 4), 0u);
  const int64_t cw = (v + 63) / 64;
  std::vector<uint64_t> colbits((size_t)(n * cw), 0ull);
  for (int64_t r = 0; r < v; ++r)
    for (int i = 0; i < n; ++i)
      if (hx[(size_t)(r * ld + i)] == 1.0f) {
        hk[(size_t)(((r >> 7) * npad + i) * 4 + ((r & 127) >> 5))] |= 1u << (r & 31);
        colbits[(size_t)(i * cw + (r >> 6))] |= 1ull << (r & 63);
      }
  std::vector<int32_t> href((size_t)n * n, 0);
  for (int i = 0; i < n; ++i)
    for (int j = i; j < n; ++j) {
      int32_t c = 0;
      for (int64_t w = 0; w < cw; ++w) c += __builtin_popcountll(colbits[(size_t)(i * cw + w)] & colbits[(size_t)(j * cw + w)]);
      href[(size_t)i * n + j] = c;
    }
  int8_t *k1, *k2, *kref;
  int32_t *s32, *flag, *sref;
  CK(hipMalloc(&k1, (size_t)kbytes));
  CK(hipMalloc(&k2, (size_t)kbytes));
  CK(hipMalloc(&kref, (size_t)kbytes));
  CK(hipMalloc(&s32, (size_t)n * n * 4));
  CK(hipMalloc(&sref, (size_t)n * n * 4));
  CK(hipMalloc(&flag, 64));
  CK(hipMemset(flag, 0, 64));
  CK(hipMemcpy(kref, hk.data(), (size_t)kbytes, hipMemcpyHostToDevice));
  CK(hipMemcpy(sref, href.data(), (size_t)n * n * 4, hipMemcpyHostToDevice));
  int bad = 0;
  auto check_k = [&](const char* what, int8_t* k) {
    const unsigned long long d = count_diff(k, kref, kbytes, cnt);
    std::printf("small n=%d v=%lld ld=%lld  operand %-28s %s (%llu words differ)\n", n, (long long)v, (long long)ld, what,
                d ? "MISMATCH" : "ok", d);
    bad += d != 0;
  };
  // fp32 pre-pass
  CK(hipMemset(k1, 0xee, (size_t)kbytes));
  CK(launch_pack_kbits(x, 0, ld, v, n, k1, flag, 0, nblk));
  check_k("pack_kbits<float>", k1);
  // uint8 pre-pass, two strides: ld (generic or 4-byte path) and a multiple of 8 (8-byte path)
  for (int64_t ld8 : {ld, (ld + 7) / 8 * 8}) {
    uint8_t* x8;
    CK(hipMalloc(&x8, (size_t)(v * ld8 + 64)));
    hipLaunchKernelGGL(to_u8_kernel, dim3(1024), dim3(256), 0, 0, x, x8, v, (int64_t)n, ld, ld8);
    CK(hipMemset(k2, 0xee, (size_t)kbytes));
    CK(launch_pack_kbits(x8, 1, ld8, v, n, k2, flag, 0, nblk));
    check_k(ld8 % 8 ? "pack_kbits<uint8>" : "pack_u8x8_kbits", k2);
    CK(hipFree(x8));
  }
  // carrier bitsets: padded stride (generic path) and a 4-word-aligned stride (vector path)
  for (int64_t ldw : {(int64_t)(n + 31) / 32 + 1, ((int64_t)(n + 31) / 32 + 3) / 4 * 4}) {
    uint32_t* bits;
    CK(hipMalloc(&bits, (size_t)(v * ldw * 4 + 64)));
    hipLaunchKernelGGL(to_bits_kernel, dim3(1024), dim3(256), 0, 0, x, bits, v, (int64_t)n, ld, ldw);
    CK(hipMemset(k2, 0xee, (size_t)kbytes));
    CK(launch_transpose_bits_kbits(bits, ldw, v, n, k2, 0, nblk));
    check_k(ldw % 4 ? "transpose_bits_kbits<1>" : "transpose_bits_kbits<4>", k2);
    CK(hipFree(bits));
  }
  {  // CSR carrier lists
    std::vector<int32_t> idx;
    std::vector<int64_t> offs((size_t)v + 1, 0);
    for (int64_t r = 0; r < v; ++r) {
      for (int i = 0; i < n; ++i)
        if (hx[(size_t)(r * ld + i)] == 1.0f) idx.push_back(i);
      offs[(size_t)r + 1] = (int64_t)idx.size();
    }
    if (idx.empty()) idx.push_back(0);
    int32_t* didx;
    int64_t* doffs;
    CK(hipMalloc(&didx, idx.size() * 4));
    CK(hipMalloc(&doffs, offs.size() * 8));
    CK(hipMemcpy(didx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(doffs, offs.data(), offs.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(k2, 0xee, (size_t)kbytes));
    CK(launch_densify_csr_kbits(didx, doffs, v, 0, k2, n, flag, 0, nblk));
    check_k("densify_csr_kbits", k2);
    CK(hipFree(didx));
    CK(hipFree(doffs));
  }
  int32_t hflag = 0;
  CK(hipMemcpy(&hflag, flag, 4, hipMemcpyDeviceToHost));
  if (hflag) { std::printf("small n=%d: flag word %d (expected 0)\n", n, hflag); ++bad; }
  // contraction, three launch modes
  for (int mode : {0, 2, 4}) {
    CK(hipMemset(s32, 0, (size_t)n * n * 4));
    hipError_t e = launch_gram_kbits(k1, v, n, s32, num_cu, 0, mode);
    if (e != hipSuccess) {
      (void)hipGetLastError();
      std::printf("small n=%d v=%lld  contraction mode %d: does not fit (%s)\n", n, (long long)v, mode, hipGetErrorString(e));
      continue;
    }
    CK(hipDeviceSynchronize());
    const unsigned long long d = count_diff(s32, sref, (int64_t)n * n * 4, cnt);
    std::printf("small n=%d v=%lld  contraction mode %d vs host popcount Gram: %s (%llu entries differ)\n", n, (long long)v, mode,
                d ? "MISMATCH" : "ok", d);
    bad += d != 0;
    if (d) {
      std::vector<int32_t> hb((size_t)n * n);
      CK(hipMemcpy(hb.data(), s32, (size_t)n * n * 4, hipMemcpyDeviceToHost));
      int by_wave[2][4] = {}, by_mi[4] = {}, by_ni[2] = {}, by_r[8] = {}, by_hi[2] = {}, shown = 0, diag = 0, offd = 0;
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
          const int32_t a = href[(size_t)i * n + j], b = hb[(size_t)i * n + j];
          if (a == b) continue;
          (i / 256 == j / 256 ? diag : offd)++;
          by_wave[(i % 256) / 128][(j % 256) / 64]++;
          by_mi[(i % 128) / 32]++;
          by_ni[(j % 64) / 32]++;
          const int ri = i % 32;
          by_hi[(ri >> 2) & 1]++;
          by_r[(ri & 3) + 4 * ((ri >> 3) & 1)]++;
          if (shown++ < 6) std::printf("   S[%d][%d]: host %d, device %d\n", i, j, a, b);
        }
      std::printf("   diag tiles %d, off-diag %d; by wave (wm, wn):", diag, offd);
      for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) std::printf(" %d", by_wave[a][b]);
      std::printf("; by mi: %d %d %d %d; by ni: %d %d; by hi: %d %d; by r&3 (+4 for odd 8-row groups):", by_mi[0], by_mi[1], by_mi[2], by_mi[3],
                  by_ni[0], by_ni[1], by_hi[0], by_hi[1]);
      for (int a = 0; a < 8; ++a) std::printf(" %d", by_r[a]);
      std::printf("\n");
    }
  }
  CK(hipFree(x)); CK(hipFree(k1)); CK(hipFree(k2)); CK(hipFree(kref)); CK(hipFree(s32)); CK(hipFree(sref)); CK(hipFree(flag));
  return bad;
}

// ---- --pipe-study (r03r): what bounds the pipelined fp32 step? ------------------------------------------------------------
//  * even split with 96 .. 160 contraction workgroups beside the pre-pass, with the duration of either kernel inside the pipeline;
//  * the same with the contraction reading an ALIASED operand: the virtual range of a whole operand buffer backed by one 2-MiB
//    physical chunk mapped over and over (hipMemMap), so that the contraction's operand stream hits in the L2s and costs no
//    fabric / HBM traffic (S is garbage, the time is what is asked for): the upper bound of what sharing operand rows between
//    workgroups could give.
static int pipe_study_main(int n, int64_t v, int reps, int num_cu) {
  const int npad = (int)gram_packed_npad(n);
  const int64_t ld = n;
  const int64_t nblk = gram_kb_pad(v, 2) / 4;
  const size_t kbytes = (size_t)nblk * npad * 16;
  float* x;
  int8_t* k1[2];
  int32_t *sb, *flag;
  CK(hipMalloc(&x, (size_t)(v * ld * 4)));
  for (int b = 0; b < 2; ++b) CK(hipMalloc(&k1[b], kbytes));
  CK(hipMalloc(&sb, (size_t)n * n * 4));
  CK(hipMalloc(&flag, 64));
  CK(hipMemset(flag, 0, 64));
  hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, x, v, (int64_t)n, ld, 777u, 0x18000000u);
  CK(launch_pack_kbits(x, 0, ld, v, n, k1[0], flag, 0, nblk));
  CK(launch_pack_kbits(x, 0, ld, v, n, k1[1], flag, 0, nblk));
  CK(hipDeviceSynchronize());
  // the aliased operand
  int8_t* ka = nullptr;
  {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    const size_t chunk = (size_t)2 << 20;
    const size_t total = (kbytes + chunk - 1) / chunk * chunk;
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, total, chunk, nullptr, 0));
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, chunk, &prop, 0));
    for (size_t off = 0; off < total; off += chunk) CK(hipMemMap((char*)va + off, chunk, 0, h, 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, total, &acc, 1));
    ka = (int8_t*)va;
    CK(hipMemcpy(ka, k1[0], chunk, hipMemcpyDeviceToDevice));
    std::printf("aliased operand: %zu MiB of virtual range on one 2-MiB chunk\n", total >> 20);
  }
  const double mv = (double)v / 1e6;
  auto line = [&](const char* what, float ms) { std::printf("time  %-66s %8.3f ms  (%.3f ms per 10^6 variants)\n", what, ms, ms / mv); };
  line("pack_kbits<float> alone", time_ms(0, reps, [&] { CK(launch_pack_kbits(x, 0, ld, v, n, k1[0], flag, 0, nblk)); }));
  for (int cus : {256, 128}) {
    std::string t = " [" + std::to_string(cus) + " workgroups]";
    line(("contraction even split, real operand" + t).c_str(), time_ms(0, reps, [&] { CK(launch_gram_kbits(k1[0], v, n, sb, cus, 0, 4)); }));
    line(("contraction even split, ALIASED operand" + t).c_str(), time_ms(0, reps, [&] { CK(launch_gram_kbits(ka, v, n, sb, cus, 0, 4)); }));
  }
  line("contraction lock-step, real operand [whole chip]", time_ms(0, reps, [&] { CK(launch_gram_kbits(k1[0], v, n, sb, num_cu, 0, 2)); }));
  line("contraction lock-step, ALIASED operand [whole chip]", time_ms(0, reps, [&] { CK(launch_gram_kbits(ka, v, n, sb, num_cu, 0, 2)); }));

  hipStream_t ps, gs;
  CK(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&gs, hipStreamNonBlocking));
  const int K = 12;
  hipEvent_t packed[2], consumed[2], pe[K][2], ge[K][2];
  for (int b = 0; b < 2; ++b) {
    CK(hipEventCreateWithFlags(&packed[b], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&consumed[b], hipEventDisableTiming));
  }
  for (int k = 0; k < K; ++k)
    for (int j = 0; j < 2; ++j) { CK(hipEventCreate(&pe[k][j])); CK(hipEventCreate(&ge[k][j])); }
  auto pipeline = [&](const char* what, int gram_mode, int gram_cus, int delay_us, bool aliased) {
    double best = 1e30, bp = 0, bg = 0;
    for (int round = 0; round < 3; ++round) {
      CK(hipDeviceSynchronize());
      const double t0 = now_ms();
      for (int k = 0; k < K; ++k) {
        const int b = k & 1;
        if (k >= 2) CK(hipStreamWaitEvent(ps, consumed[b], 0));
        if (k >= 1 && delay_us > 0) hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, ps, (long long)delay_us * 100);
        CK(hipEventRecord(pe[k][0], ps));
        CK(launch_pack_kbits(x, 0, ld, v, n, k1[b], flag, ps, nblk));
        CK(hipEventRecord(pe[k][1], ps));
        CK(hipEventRecord(packed[b], ps));
        CK(hipStreamWaitEvent(gs, packed[b], 0));
        CK(hipEventRecord(ge[k][0], gs));
        CK(launch_gram_kbits(aliased ? ka : k1[b], v, n, sb, gram_cus, gs, gram_mode));
        CK(hipEventRecord(ge[k][1], gs));
        CK(hipEventRecord(consumed[b], gs));
      }
      CK(hipDeviceSynchronize());
      const double t = (now_ms() - t0) / K;
      double sp = 0, sg = 0;
      for (int k = 2; k < K - 1; ++k) {  // steady state
        float a = 0, c = 0;
        CK(hipEventElapsedTime(&a, pe[k][0], pe[k][1]));
        CK(hipEventElapsedTime(&c, ge[k][0], ge[k][1]));
        sp += a; sg += c;
      }
      if (t < best) { best = t; bp = sp / (K - 3); bg = sg / (K - 3); }
    }
    std::printf("pipe  %-58s %8.3f ms per step (%.0f M variants/s); inside: pre-pass %.3f ms, contraction %.3f ms\n", what, best,
                v / best / 1e3, bp, bg);
  };
  for (int cus : {96, 104, 112, 120, 128, 136, 144, 160}) {
    std::string t = "even split, " + std::to_string(cus) + " workgroups";
    pipeline(t.c_str(), 4, cus, 10, false);
  }
  for (int cus : {112, 128, 144}) {
    std::string t = "even split, " + std::to_string(cus) + " workgroups, ALIASED operand";
    pipeline(t.c_str(), 4, cus, 10, true);
  }
  pipeline("lock-step on half the chip (110)", 2, num_cu / 2, 10, false);
  pipeline("lock-step on half the chip (110), ALIASED operand", 2, num_cu / 2, 10, true);
  pipeline("even split 128, head start 20 us", 4, 128, 20, false);
  pipeline("even split 128, head start 5 us", 4, 128, 5, false);
  return 0;
}

// ---- --coreside (r03s): pre-pass and contraction on the SAME CUs ----------------------------------------------------------------
// The contraction held to 224 VGPRs per wave (gram_kbits_capped_kernel) leaves every SIMD 64 registers: room for one wave of the
// persistent LDS-DMA-ring pre-pass (pack_kbits_ring_kernel, <= 64 VGPRs, 64 KiB of LDS).  Contraction first (one workgroup per
// CU, even split over the whole chip), then one ring workgroup lands on every CU beside it.
static int coreside_main(int n, int64_t v, int reps, int num_cu, unsigned long long* cnt) {
  const int npad = (int)gram_packed_npad(n);
  const int64_t ld = n;
  const int64_t nblk = gram_kb_pad(v, 2) / 4;
  const size_t kbytes = (size_t)nblk * npad * 16;
  float* x;
  int8_t* k1[2];
  int32_t *sa, *sb, *flag;
  CK(hipMalloc(&x, (size_t)(v * ld * 4)));
  for (int b = 0; b < 2; ++b) CK(hipMalloc(&k1[b], kbytes));
  CK(hipMalloc(&sa, (size_t)n * n * 4));
  CK(hipMalloc(&sb, (size_t)n * n * 4));
  CK(hipMalloc(&flag, 64));
  CK(hipMemset(flag, 0, 64));
  hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, x, v, (int64_t)n, ld, 777u, 0x18000000u);
  CK(launch_pack_kbits(x, 0, ld, v, n, k1[0], flag, 0, nblk));
  CK(hipDeviceSynchronize());
  int bad = 0;
  for (int ring : {108, 116, 8, 5016, 5018, 5001}) {
    CK(hipMemset(k1[1], 0xa5, kbytes));
    CK(launch_pack_kbits_ring(x, ld, v, n, k1[1], flag, 0, nblk, num_cu, ring));
    CK(hipDeviceSynchronize());
    const unsigned long long d = count_diff(k1[0], k1[1], (int64_t)kbytes, cnt);
    std::printf("ring pre-pass (ring %d) operand vs pack_kbits operand: %s (%llu words differ)\n", ring, d ? "MISMATCH" : "identical", d);
    bad += d != 0;
  }
  {  // ragged small shape through the ring kernel
    const int n2 = 1000;
    const int64_t v2 = 4100, ld2 = 1000;
    const int npad2 = (int)gram_packed_npad(n2);
    const int64_t nblk2 = gram_kb_pad(v2, 2) / 4;
    float* x2;
    int8_t *ka, *kb;
    CK(hipMalloc(&x2, (size_t)(v2 * ld2 * 4)));
    CK(hipMalloc(&ka, (size_t)nblk2 * npad2 * 16));
    CK(hipMalloc(&kb, (size_t)nblk2 * npad2 * 16));
    hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, x2, v2, (int64_t)n2, ld2, 99u, 0x30000000u);
    CK(launch_pack_kbits(x2, 0, ld2, v2, n2, ka, flag, 0, nblk2));
    CK(hipMemset(kb, 0x5a, (size_t)nblk2 * npad2 * 16));
    CK(launch_pack_kbits_ring(x2, ld2, v2, n2, kb, flag, 0, nblk2, 7, 108));
    CK(hipDeviceSynchronize());
    const unsigned long long d = count_diff(ka, kb, nblk2 * npad2 * 16, cnt);
    std::printf("ring pre-pass, n=1000 v=4100, 7 workgroups: %s (%llu words differ)\n", d ? "MISMATCH" : "identical", d);
    bad += d != 0;
    CK(hipFree(x2)); CK(hipFree(ka)); CK(hipFree(kb));
  }
  g_kbits_variant = 0;
  CK(hipMemset(sa, 0, (size_t)n * n * 4));
  CK(launch_gram_kbits(k1[0], v, n, sa, num_cu, 0, 4));
  {
    CK(launch_pack_kbits_ring(x, ld, v, n, k1[1], flag, 0, nblk, 2 * num_cu, 108));
    CK(hipMemset(sb, 0, (size_t)n * n * 4));
    CK(launch_gram_kbits(k1[1], v, n, sb, num_cu, 0, 2));
    CK(hipDeviceSynchronize());
    const unsigned long long d = count_diff(sa, sb, (int64_t)n * n * 4, cnt);
    std::printf("S(ring pre-pass operand, lock-step) vs S(pack_kbits operand, even split): %s (%llu entries differ)\n", d ? "MISMATCH" : "bit-identical", d);
    bad += d != 0;
  }
  CK(launch_pack_kbits(x, 0, ld, v, n, k1[1], flag, 0, nblk));
  const char* vname[9] = {"shipped (256 VGPRs, ring 3)", "", "", "ring 4", "", "capped 224, ring 3", "capped 224, ring 4", "capped 224, ring 6", "ring 6"};
  for (int var : {5, 6, 7, 8}) {
    g_kbits_variant = var;
    CK(hipMemset(sb, 0, (size_t)n * n * 4));
    CK(launch_gram_kbits(k1[0], v, n, sb, num_cu, 0, 4));
    CK(hipDeviceSynchronize());
    const unsigned long long d = count_diff(sa, sb, (int64_t)n * n * 4, cnt);
    std::printf("S(%s) vs S(shipped): %s (%llu entries differ)\n", vname[var], d ? "MISMATCH" : "bit-identical", d);
    bad += d != 0;
  }
  g_kbits_variant = 0;
  const double mv = (double)v / 1e6;
  auto line = [&](const char* what, float ms) { std::printf("time  %-66s %8.3f ms  (%.3f ms per 10^6 variants)\n", what, ms, ms / mv); };
  line("pack_kbits<float> (shipped) alone", time_ms(0, reps, [&] { CK(launch_pack_kbits(x, 0, ld, v, n, k1[1], flag, 0, nblk)); }));
  for (int ring : {108, 5000, 5016, 5001})
    line(("ring pre-pass alone, 256 workgroups, ring " + std::to_string(ring)).c_str(),
         time_ms(0, reps, [&] { CK(launch_pack_kbits_ring(x, ld, v, n, k1[1], flag, 0, nblk, num_cu, ring)); }));
  line("ring pre-pass alone, 512 workgroups, ring 108", time_ms(0, reps, [&] { CK(launch_pack_kbits_ring(x, ld, v, n, k1[1], flag, 0, nblk, 2 * num_cu, 108)); }));
  line("ring pre-pass alone, 512 workgroups, ring 116", time_ms(0, reps, [&] { CK(launch_pack_kbits_ring(x, ld, v, n, k1[1], flag, 0, nblk, 2 * num_cu, 116)); }));
  for (int var : {0, 3, 5, 6, 7, 8}) {
    g_kbits_variant = var;
    line((std::string("contraction alone, even split 256: ") + vname[var]).c_str(), time_ms(0, reps, [&] { CK(launch_gram_kbits(k1[0], v, n, sb, num_cu, 0, 4)); }));
  }
  g_kbits_variant = 0;

  hipStream_t ps, gs;
  CK(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&gs, hipStreamNonBlocking));
  const int K = 12;
  hipEvent_t packed[2], consumed[2], pe[K][2], ge[K][2];
  for (int b = 0; b < 2; ++b) {
    CK(hipEventCreateWithFlags(&packed[b], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&consumed[b], hipEventDisableTiming));
  }
  for (int k = 0; k < K; ++k)
    for (int j = 0; j < 2; ++j) { CK(hipEventCreate(&pe[k][j])); CK(hipEventCreate(&ge[k][j])); }
  // ring < 0: the shipped pre-pass; gram_cus workgroups of contraction variant `var`
  int8_t* ka = nullptr;
  {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    const size_t chunk = (size_t)2 << 20;
    const size_t total = (kbytes + chunk - 1) / chunk * chunk;
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, total, chunk, nullptr, 0));
    hipMemGenericAllocationHandle_t h;
    CK(hipMemCreate(&h, chunk, &prop, 0));
    for (size_t off = 0; off < total; off += chunk) CK(hipMemMap((char*)va + off, chunk, 0, h, 0));
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, total, &acc, 1));
    ka = (int8_t*)va;
    CK(hipMemcpy(ka, k1[0], chunk, hipMemcpyDeviceToDevice));
  }
  int gram_mode = 4;
  bool aliased = false;
  auto pipeline = [&](const char* what, int var, int gram_cus, int ring, int ring_wgs, int delay_us) {
    double best = 1e30, bp = 0, bg = 0;
    for (int round = 0; round < 3; ++round) {
      CK(hipDeviceSynchronize());
      const double t0 = now_ms();
      for (int k = 0; k < K; ++k) {
        const int b = k & 1;
        if (k >= 2) CK(hipStreamWaitEvent(ps, consumed[b], 0));
        if (k >= 1 && delay_us > 0) hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, ps, (long long)delay_us * 100);
        CK(hipEventRecord(pe[k][0], ps));
        if (ring < 0) CK(launch_pack_kbits(x, 0, ld, v, n, k1[b], flag, ps, nblk));
        else CK(launch_pack_kbits_ring(x, ld, v, n, k1[b], flag, ps, nblk, ring_wgs, ring));
        CK(hipEventRecord(pe[k][1], ps));
        CK(hipEventRecord(packed[b], ps));
        CK(hipStreamWaitEvent(gs, packed[b], 0));
        CK(hipEventRecord(ge[k][0], gs));
        g_kbits_variant = var;
        CK(launch_gram_kbits(aliased ? ka : k1[b], v, n, sb, gram_cus, gs, gram_mode));
        g_kbits_variant = 0;
        CK(hipEventRecord(ge[k][1], gs));
        CK(hipEventRecord(consumed[b], gs));
      }
      CK(hipDeviceSynchronize());
      const double t = (now_ms() - t0) / K;
      double sp = 0, sg = 0;
      for (int k = 2; k < K - 1; ++k) {
        float a = 0, c = 0;
        CK(hipEventElapsedTime(&a, pe[k][0], pe[k][1]));
        CK(hipEventElapsedTime(&c, ge[k][0], ge[k][1]));
        sp += a; sg += c;
      }
      if (t < best) { best = t; bp = sp / (K - 3); bg = sg / (K - 3); }
    }
    std::printf("pipe  %-74s %8.3f ms per step (%.0f M variants/s); inside: pre-pass %.3f ms, contraction %.3f ms\n", what, best,
                v / best / 1e3, bp, bg);
  };
  pipeline("shipped: pack_kbits || even split 128 (disjoint CUs)", 0, 128, -1, 0, 10);
  struct Cfg { const char* what; int var, mode, cus, ring, wgs; };
  const Cfg cfgs[] = {
      {"CO-RESIDENT: ring R8 aux 0 (default policy), 512 wgs || lock-step 220", 5, 2, 256, 5000, 512},
      {"CO-RESIDENT: ring R8 aux 2 (nt), 512 wgs || lock-step 220", 5, 2, 256, 108, 512},
      {"CO-RESIDENT: ring R8 aux 1 (sc0), 512 wgs || lock-step 220", 5, 2, 256, 5001, 512},
      {"CO-RESIDENT: ring R8 aux 3 (sc0 nt), 512 wgs || lock-step 220", 5, 2, 256, 5003, 512},
      {"CO-RESIDENT: ring R8 aux 16 (sc1), 512 wgs || lock-step 220", 5, 2, 256, 5016, 512},
      {"CO-RESIDENT: ring R8 aux 17 (sc0 sc1), 512 wgs || lock-step 220", 5, 2, 256, 5017, 512},
      {"CO-RESIDENT: ring R8 aux 18 (sc1 nt), 512 wgs || lock-step 220", 5, 2, 256, 5018, 512},
      {"CO-RESIDENT: ring R8 aux 0 (default policy), 512 wgs || lock-step 220 (again)", 5, 2, 256, 5000, 512},
      {"CO-RESIDENT: ring R8 aux 0, 512 wgs || even split 256", 5, 4, 256, 5000, 512},
      {"CO-RESIDENT: ring R8 aux 0, 768 wgs || lock-step 220", 5, 2, 256, 5000, 768},
  };
  for (const Cfg& c : cfgs) {
    gram_mode = c.mode;
    pipeline(c.what, c.var, c.cus, c.ring, c.wgs, 10);
  }
  gram_mode = 4;
  std::printf("%s\n", bad ? "RESULT: FAILED" : "RESULT: ok");
  return bad ? 1 : 0;
}

// ---- --aln (r04x / r04y): the ring pre-pass over row pitches on and off the 128-byte lines, alone and co-resident ---------------
// (r04x also had an "aligned window" form of the kernel for rows off the lines: per row class r mod 4 the wave read the aligned
// 1 KiB its group starts in and permuted the variants of a 128-block accordingly.  Bit-exact, and worth nothing: 1.920 vs 1.927 ms
// at ld = 2504.  The pitch sweep shows why -- 2520 (off the lines) runs like 2528, 2592 (on them) like 2504, and the same 2504
// tile allocated twice differs by 6 %: the spread is where the 10 GB tile happens to lie, not how its rows sit in their lines.
// profiles/r04y_ring_pitch_sweep.txt.  The form was removed.)
static std::vector<int64_t> g_aln_lds;
static bool g_aln_alone = false;
static int aln_main(int n, int64_t v, int reps, int num_cu, unsigned long long* cnt) {
  const int npad = (int)gram_packed_npad(n);
  const int64_t nblk = gram_kb_pad(v, 2) / 4;
  const size_t kbytes = (size_t)nblk * npad * 16;
  int8_t* k1[2];
  int32_t *sa, *sb, *flag;
  for (int b = 0; b < 2; ++b) CK(hipMalloc(&k1[b], kbytes));
  CK(hipMalloc(&sa, (size_t)n * n * 4));
  CK(hipMalloc(&sb, (size_t)n * n * 4));
  CK(hipMalloc(&flag, 64));
  CK(hipMemset(flag, 0, 64));
  hipStream_t ps, gs;
  CK(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&gs, hipStreamNonBlocking));
  const int K = 12;
  hipEvent_t packed[2], consumed[2], pe[K][2], ge[K][2];
  for (int b = 0; b < 2; ++b) {
    CK(hipEventCreateWithFlags(&packed[b], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&consumed[b], hipEventDisableTiming));
  }
  for (int k = 0; k < K; ++k)
    for (int j = 0; j < 2; ++j) { CK(hipEventCreate(&pe[k][j])); CK(hipEventCreate(&ge[k][j])); }
  int bad = 0;
  const double mv = (double)v / 1e6;
  if (g_aln_lds.empty()) g_aln_lds = {(int64_t)n, (int64_t)((n + 31) / 32 * 32), (int64_t)n};
  for (int64_t ld : g_aln_lds) {
    float* x;
    CK(hipMalloc(&x, (size_t)(v * ld * 4)));
    hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, x, v, (int64_t)n, ld, 777u, 0x18000000u);
    CK(hipDeviceSynchronize());
    std::printf("-- ld = %lld floats (%lld B per row, %s)\n", (long long)ld, (long long)ld * 4,
                (ld * 4) % 128 ? "rows OFF the 128-byte lines" : "rows on 128-byte lines");
    // S through both forms
    for (int form = 0; form < 2; ++form) {
      CK(launch_pack_kbits_ring(x, ld, v, n, k1[form], flag, 0, nblk, 2 * num_cu, 8));
      CK(hipMemset(form ? sb : sa, 0, (size_t)n * n * 4));
      CK(launch_gram_kbits(k1[form], v, n, form ? sb : sa, num_cu, 0, 4));
    }
    CK(hipDeviceSynchronize());
    const unsigned long long d = count_diff(sa, sb, (int64_t)n * n * 4, cnt);
    std::printf("S(ring pre-pass twice): %s (%llu entries differ)\n", d ? "MISMATCH" : "bit-identical", d);
    bad += d != 0;
    auto line = [&](const char* what, float ms) { std::printf("time  %-66s %8.3f ms  (%.3f ms per 10^6 variants)\n", what, ms, ms / mv); };
    for (int wgs : {num_cu, 2 * num_cu})
      for (int ring : {8, 8})
        line((std::string("ring alone, ") + std::to_string(wgs) + " workgroups").c_str(),
             time_ms(0, reps, [&] { CK(launch_pack_kbits_ring(x, ld, v, n, k1[1], flag, 0, nblk, wgs, ring)); }));
    for (int ring : {8, 8}) {
      if (g_aln_alone) break;
      double best = 1e30, bp = 0, bg = 0;
      for (int round = 0; round < 3; ++round) {
        CK(hipDeviceSynchronize());
        const double t0 = now_ms();
        for (int k = 0; k < K; ++k) {
          const int b = k & 1;
          if (k >= 2) CK(hipStreamWaitEvent(ps, consumed[b], 0));
          if (k >= 1) hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, ps, (long long)10 * 100);
          CK(hipEventRecord(pe[k][0], ps));
          CK(launch_pack_kbits_ring(x, ld, v, n, k1[b], flag, ps, nblk, 2 * num_cu, ring));
          CK(hipEventRecord(pe[k][1], ps));
          CK(hipEventRecord(packed[b], ps));
          CK(hipStreamWaitEvent(gs, packed[b], 0));
          CK(hipEventRecord(ge[k][0], gs));
          g_kbits_variant = 5;
          CK(launch_gram_kbits(k1[b], v, n, sb, num_cu, gs, 2));
          g_kbits_variant = 0;
          CK(hipEventRecord(ge[k][1], gs));
          CK(hipEventRecord(consumed[b], gs));
        }
        CK(hipDeviceSynchronize());
        const double t = (now_ms() - t0) / K;
        double sp = 0, sg = 0;
        for (int k = 2; k < K - 1; ++k) {
          float a = 0, c = 0;
          CK(hipEventElapsedTime(&a, pe[k][0], pe[k][1]));
          CK(hipEventElapsedTime(&c, ge[k][0], ge[k][1]));
          sp += a; sg += c;
        }
        if (t < best) { best = t; bp = sp / (K - 3); bg = sg / (K - 3); }
      }
      std::printf("pipe  CO-RESIDENT %-28s 512 wgs || lock-step: %8.3f ms per step (%.0f M variants/s); pre-pass %.3f, contraction %.3f\n",
                  "ring pre-pass", best, v / best / 1e3, bp, bg);
    }
    CK(hipFree(x));
  }
  std::printf("%s\n", bad ? "RESULT: FAILED" : "RESULT: ok");
  return bad ? 1 : 0;
}

// ---- --coreside-alt (r03zd): the uint8 and bitset boundaries with their pre-pass beside the contraction -------------------------
static int u8_ring_case(int n, int64_t v, int64_t ld8, uint32_t thr, int wgs, unsigned long long* cnt, int32_t* flag) {
  const int npad = (int)gram_packed_npad(n);
  const int64_t nblk = gram_kb_pad(v, 2) / 4;
  const size_t kbytes = (size_t)nblk * npad * 16;
  float* xf;
  uint8_t* x8;
  int8_t *ka, *kb;
  CK(hipMalloc(&xf, (size_t)(v * n * 4)));
  CK(hipMalloc(&x8, (size_t)(v * ld8)));   // exactly the tile: a read beyond its last row is out of the allocation
  CK(hipMalloc(&ka, kbytes));
  CK(hipMalloc(&kb, kbytes));
  hipLaunchKernelGGL(fill_kernel, dim3(1024), dim3(256), 0, 0, xf, v, (int64_t)n, (int64_t)n, 4242u + (uint32_t)n, thr);
  hipLaunchKernelGGL(to_u8_kernel, dim3(1024), dim3(256), 0, 0, xf, x8, v, (int64_t)n, (int64_t)n, ld8);
  CK(launch_pack_kbits(x8, 1, ld8, v, n, ka, flag, 0, nblk));
  CK(hipMemset(kb, 0x5a, kbytes));
  CK(launch_pack_kbits_ring_u8(x8, ld8, v, n, kb, flag, 0, nblk, wgs, (n & 1) ? 8 : 108));
  CK(hipDeviceSynchronize());
  const unsigned long long d = count_diff(ka, kb, (int64_t)kbytes, cnt);
  std::printf("u8 ring pre-pass n=%d v=%lld ld=%lld (%d workgroups) vs pack_kbits<uint8>: %s (%llu words differ)\n", n, (long long)v,
              (long long)ld8, wgs, d ? "MISMATCH" : "identical", d);
  CK(hipFree(xf)); CK(hipFree(x8)); CK(hipFree(ka)); CK(hipFree(kb));
  return d != 0;
}

static int coreside_alt_main(int n, int64_t v, int reps, int num_cu, unsigned long long* cnt) {
  const int npad = (int)gram_packed_npad(n);
  const int64_t ld = n;
  const int64_t nblk = gram_kb_pad(v, 2) / 4;
  const size_t kbytes = (size_t)nblk * npad * 16;
  int32_t *sa, *sb, *flag;
  CK(hipMalloc(&flag, 64));
  CK(hipMemset(flag, 0, 64));
  int bad = 0;
  bad += u8_ring_case(1000, 4100, 1000, 0x30000000u, 7, cnt, flag);     // ld % 16 == 8: the last lane of a row is shifted
  bad += u8_ring_case(257, 129, 264, 0xc0000000u, 3, cnt, flag);
  bad += u8_ring_case(2504, 3000, 2512, 0x20000000u, 64, cnt, flag);    // ld % 16 == 0, padding columns
  bad += u8_ring_case(1025, 777, 1032, 0x20000000u, 5, cnt, flag);
  bad += u8_ring_case(33, 1000, 40, 0x60000000u, 2, cnt, flag);
  int32_t hflag = 0;
  CK(hipMemcpy(&hflag, flag, 4, hipMemcpyDeviceToHost));
  std::printf("flag word after the binary cases: %d\n", hflag);
  float* x;
  int8_t* k1[2];
  CK(hipMalloc(&x, (size_t)(v * ld * 4)));
  for (int b = 0; b < 2; ++b) CK(hipMalloc(&k1[b], kbytes));
  CK(hipMalloc(&sa, (size_t)n * n * 4));
  CK(hipMalloc(&sb, (size_t)n * n * 4));
  hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, x, v, (int64_t)n, ld, 777u, 0x18000000u);
  const int64_t ld8 = (n + 7) / 8 * 8;
  uint8_t* x8;
  CK(hipMalloc(&x8, (size_t)(v * ld8)));
  hipLaunchKernelGGL(to_u8_kernel, dim3(8192), dim3(256), 0, 0, x, x8, v, (int64_t)n, ld, ld8);
  const int64_t ldw = ((n + 31) / 32 + 3) / 4 * 4;
  uint32_t* bits;
  CK(hipMalloc(&bits, (size_t)(v * ldw * 4)));
  hipLaunchKernelGGL(to_bits_kernel, dim3(8192), dim3(256), 0, 0, x, bits, v, (int64_t)n, ld, ldw);
  CK(launch_pack_kbits(x8, 1, ld8, v, n, k1[0], flag, 0, nblk));
  CK(hipMemset(k1[1], 0x33, kbytes));
  CK(launch_pack_kbits_ring_u8(x8, ld8, v, n, k1[1], flag, 0, nblk, num_cu, 8));
  CK(hipDeviceSynchronize());
  {
    const unsigned long long d = count_diff(k1[0], k1[1], (int64_t)kbytes, cnt);
    std::printf("full size: u8 ring operand vs pack_u8x8_kbits operand: %s (%llu words differ)\n", d ? "MISMATCH" : "identical", d);
    bad += d != 0;
  }
  {  // a multiplicity must raise the flag
    uint8_t two = 2;
    CK(hipMemcpy(x8 + 12345 * ld8 + 77, &two, 1, hipMemcpyHostToDevice));
    CK(hipMemset(flag, 0, 64));
    CK(launch_pack_kbits_ring_u8(x8, ld8, v, n, k1[1], flag, 0, nblk, num_cu, 8));
    CK(hipMemcpy(&hflag, flag, 4, hipMemcpyDeviceToHost));
    std::printf("a byte of value 2 in the tile: flag word %d (%s)\n", hflag, (hflag & 8) ? "raised" : "NOT RAISED");
    bad += (hflag & 8) == 0;
    uint8_t back = 0;  // restore from the fp32 source
    float f = 0;
    CK(hipMemcpy(&f, x + 12345 * ld + 77, 4, hipMemcpyDeviceToHost));
    back = (uint8_t)f;
    CK(hipMemcpy(x8 + 12345 * ld8 + 77, &back, 1, hipMemcpyHostToDevice));
    CK(hipMemset(flag, 0, 64));
  }
  const double mv = (double)v / 1e6;
  auto line = [&](const char* what, float ms) { std::printf("time  %-66s %8.3f ms  (%.3f ms per 10^6 variants)\n", what, ms, ms / mv); };
  line("pack_u8x8_kbits (shipped) alone", time_ms(0, reps, [&] { CK(launch_pack_kbits(x8, 1, ld8, v, n, k1[1], flag, 0, nblk)); }));
  line("u8 ring pre-pass alone, 256 workgroups", time_ms(0, reps, [&] { CK(launch_pack_kbits_ring_u8(x8, ld8, v, n, k1[1], flag, 0, nblk, num_cu, 8)); }));
  line("u8 ring pre-pass alone, 512 workgroups", time_ms(0, reps, [&] { CK(launch_pack_kbits_ring_u8(x8, ld8, v, n, k1[1], flag, 0, nblk, 2 * num_cu, 8)); }));
  {  // the same tile with rows padded to a multiple of 16 bytes: is it the alignment of the 16-byte lane accesses?
    const int64_t ld16 = (n + 15) / 16 * 16;
    uint8_t* x16;
    CK(hipMalloc(&x16, (size_t)(v * ld16)));
    hipLaunchKernelGGL(to_u8_kernel, dim3(8192), dim3(256), 0, 0, x, x16, v, (int64_t)n, ld, ld16);
    line("u8 ring pre-pass alone, 256 workgroups, ld % 16 == 0", time_ms(0, reps, [&] { CK(launch_pack_kbits_ring_u8(x16, ld16, v, n, k1[1], flag, 0, nblk, num_cu, 8)); }));
    line("u8 ring pre-pass alone, 512 workgroups, ld % 16 == 0", time_ms(0, reps, [&] { CK(launch_pack_kbits_ring_u8(x16, ld16, v, n, k1[1], flag, 0, nblk, 2 * num_cu, 8)); }));
    line("pack_u8x8_kbits alone, ld % 16 == 0", time_ms(0, reps, [&] { CK(launch_pack_kbits(x16, 1, ld16, v, n, k1[1], flag, 0, nblk)); }));
    CK(hipFree(x16));
  }
  line("transpose_bits_kbits alone", time_ms(0, reps, [&] { CK(launch_transpose_bits_kbits(bits, ldw, v, n, k1[1], 0, nblk)); }));
  line("contraction alone, even split 256", time_ms(0, reps, [&] { CK(launch_gram_kbits(k1[0], v, n, sb, num_cu, 0, 4)); }));
  line("contraction alone, lock-step 220", time_ms(0, reps, [&] { CK(launch_gram_kbits(k1[0], v, n, sb, num_cu, 0, 2)); }));
  line("serial step u8 (pack_u8x8 + even split, one stream)", time_ms(0, reps, [&] {
         CK(launch_pack_kbits(x8, 1, ld8, v, n, k1[0], flag, 0, nblk));
         CK(launch_gram_kbits(k1[0], v, n, sb, num_cu, 0, 4));
       }));
  line("serial step bits (transpose + even split, one stream)", time_ms(0, reps, [&] {
         CK(launch_transpose_bits_kbits(bits, ldw, v, n, k1[0], 0, nblk));
         CK(launch_gram_kbits(k1[0], v, n, sb, num_cu, 0, 4));
       }));
  hipStream_t ps, gs;
  CK(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&gs, hipStreamNonBlocking));
  const int K = 12;
  hipEvent_t packed[2], consumed[2], pe[K][2], ge[K][2];
  for (int b = 0; b < 2; ++b) {
    CK(hipEventCreateWithFlags(&packed[b], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&consumed[b], hipEventDisableTiming));
  }
  for (int k = 0; k < K; ++k)
    for (int j = 0; j < 2; ++j) { CK(hipEventCreate(&pe[k][j])); CK(hipEventCreate(&ge[k][j])); }
  // kind: 2 = pack_u8x8 (shipped), 3 = u8 ring with `wgs` workgroups, 4 = bitset transpose
  auto pipeline = [&](const char* what, int kind, int wgs, int gram_mode, int gram_cus, int delay_us) {
    double best = 1e30, bp = 0, bg = 0;
    for (int round = 0; round < 3; ++round) {
      CK(hipDeviceSynchronize());
      const double t0 = now_ms();
      for (int k = 0; k < K; ++k) {
        const int b = k & 1;
        if (k >= 2) CK(hipStreamWaitEvent(ps, consumed[b], 0));
        if (k >= 1 && delay_us > 0) hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, ps, (long long)delay_us * 100);
        CK(hipEventRecord(pe[k][0], ps));
        if (kind == 2) CK(launch_pack_kbits(x8, 1, ld8, v, n, k1[b], flag, ps, nblk));
        else if (kind == 3) CK(launch_pack_kbits_ring_u8(x8, ld8, v, n, k1[b], flag, ps, nblk, wgs, 108));
        else if (kind == 5) CK(launch_pack_kbits_ring_u8(x8, ld8, v, n, k1[b], flag, ps, nblk, wgs, 8));
        else CK(launch_transpose_bits_kbits(bits, ldw, v, n, k1[b], ps, nblk));
        CK(hipEventRecord(pe[k][1], ps));
        CK(hipEventRecord(packed[b], ps));
        CK(hipStreamWaitEvent(gs, packed[b], 0));
        CK(hipEventRecord(ge[k][0], gs));
        CK(launch_gram_kbits(k1[b], v, n, sb, gram_cus, gs, gram_mode));
        CK(hipEventRecord(ge[k][1], gs));
        CK(hipEventRecord(consumed[b], gs));
      }
      CK(hipDeviceSynchronize());
      const double t = (now_ms() - t0) / K;
      double sp = 0, sg = 0;
      for (int k = 2; k < K - 1; ++k) {
        float a = 0, c = 0;
        CK(hipEventElapsedTime(&a, pe[k][0], pe[k][1]));
        CK(hipEventElapsedTime(&c, ge[k][0], ge[k][1]));
        sp += a; sg += c;
      }
      if (t < best) { best = t; bp = sp / (K - 3); bg = sg / (K - 3); }
    }
    std::printf("pipe  %-74s %8.3f ms per step (%.0f M variants/s); inside: pre-pass %.3f ms, contraction %.3f ms\n", what, best,
                v / best / 1e3, bp, bg);
  };
  pipeline("u8: pack_u8x8 (184 VGPRs, cannot share a CU) || even split 256", 2, 0, 4, num_cu, 10);
  pipeline("u8: pack_u8x8 || lock-step 220 (36 CUs free)", 2, 0, 2, num_cu, 10);
  pipeline("u8 CO-RESIDENT: u8 ring nt, 256 wgs || lock-step 220", 3, num_cu, 2, num_cu, 10);
  pipeline("u8 CO-RESIDENT: u8 ring default policy, 256 wgs || lock-step 220", 5, num_cu, 2, num_cu, 10);
  pipeline("u8 CO-RESIDENT: u8 ring nt, 256 wgs || lock-step 220 (again)", 3, num_cu, 2, num_cu, 10);
  pipeline("u8 CO-RESIDENT: u8 ring default policy, 256 wgs || lock-step 220 (again)", 5, num_cu, 2, num_cu, 10);
  pipeline("u8 CO-RESIDENT: u8 ring default policy, 256 wgs || even split 256", 5, num_cu, 4, num_cu, 10);
  pipeline("u8 CO-RESIDENT: u8 ring, 256 wgs || even split 256", 3, num_cu, 4, num_cu, 10);
  pipeline("u8 CO-RESIDENT: u8 ring, 128 wgs || even split 256", 3, num_cu / 2, 4, num_cu, 10);
  pipeline("bits CO-RESIDENT: transpose (68 VGPRs) || lock-step 220", 4, 0, 2, num_cu, 10);
  pipeline("bits CO-RESIDENT: transpose || even split 256", 4, 0, 4, num_cu, 10);
  pipeline("bits CO-RESIDENT: transpose || even split 256, no head start", 4, 0, 4, num_cu, 0);
  // S through the co-resident u8 pipeline == S of the serial path
  CK(hipMemset(sa, 0, (size_t)n * n * 4));
  CK(launch_pack_kbits(x8, 1, ld8, v, n, k1[0], flag, 0, nblk));
  CK(launch_gram_kbits(k1[0], v, n, sa, num_cu, 0, 4));
  CK(hipMemset(sb, 0, (size_t)n * n * 4));
  CK(launch_pack_kbits_ring_u8(x8, ld8, v, n, k1[1], flag, 0, nblk, num_cu, 8));
  CK(launch_gram_kbits(k1[1], v, n, sb, num_cu, 0, 2));
  CK(hipDeviceSynchronize());
  {
    const unsigned long long d = count_diff(sa, sb, (int64_t)n * n * 4, cnt);
    std::printf("S(u8 ring, lock-step) vs S(pack_u8x8, even split): %s (%llu entries differ)\n", d ? "MISMATCH" : "bit-identical", d);
    bad += d != 0;
  }
  std::printf("%s\n", bad ? "RESULT: FAILED" : "RESULT: ok");
  return bad ? 1 : 0;
}

int main(int argc, char** argv) {
  int n = 2504, reps = 5;
  bool pipe_study = false, coreside = false, coreside_alt = false, aln = false;
  int64_t v = (int64_t)1 << 20;
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--n") && i + 1 < argc) n = std::atoi(argv[++i]);
    if (!std::strcmp(argv[i], "--v") && i + 1 < argc) v = std::atoll(argv[++i]);
    if (!std::strcmp(argv[i], "--reps") && i + 1 < argc) reps = std::atoi(argv[++i]);
    if (!std::strcmp(argv[i], "--pipe-study")) pipe_study = true;
    if (!std::strcmp(argv[i], "--coreside")) coreside = true;
    if (!std::strcmp(argv[i], "--coreside-alt")) coreside_alt = true;
    if (!std::strcmp(argv[i], "--aln")) aln = true;
    if (!std::strcmp(argv[i], "--alone")) g_aln_alone = true;
    if (!std::strcmp(argv[i], "--lds") && i + 1 < argc) {
      for (char* t = std::strtok(argv[++i], ","); t; t = std::strtok(nullptr, ",")) g_aln_lds.push_back(std::atoll(t));
    }
  }
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int num_cu = prop.multiProcessorCount;
  std::printf("device %s, %d CUs\n", prop.name, num_cu);
  unsigned long long* cnt;
  CK(hipMalloc(&cnt, 8));

  if (pipe_study) return pipe_study_main(n, v, reps, num_cu);
  if (coreside) return coreside_main(n, v, reps, num_cu, cnt);
  if (coreside_alt) return coreside_alt_main(n, v, reps, num_cu, cnt);
  if (aln) return aln_main(n, v, reps, num_cu, cnt);
  int bad = 0;
  bad += small_case(1000, 777, 1003, 0x30000000u, num_cu, cnt);   // odd stride: generic paths
  bad += small_case(1000, 4100, 1000, 0x08000000u, num_cu, cnt);  // vector paths, several blocks
  bad += small_case(257, 129, 260, 0xc0000000u, num_cu, cnt);     // tile edge + dense
  bad += small_case(2504, 3000, 2504, 0x20000000u, num_cu, cnt);  // BASELINE sample count, lock-step fits
  std::printf("part 1: %s\n", bad ? "FAILED" : "all ok");

  // ---- part 2 / 3: BASELINE configs[1] size ----------------------------------------------------------------------------
  const int npad = (int)gram_packed_npad(n);
  const int64_t ld = n;
  const int64_t nkb = gram_kb_pad(v, 1);
  const int64_t nblk = gram_kb_pad(v, 2) / 4;
  float* x;
  int8_t *p4[2], *k1[2];
  int32_t *sa, *sb, *flag;
  CK(hipMalloc(&x, (size_t)(v * ld * 4)));
  for (int b = 0; b < 2; ++b) {
    CK(hipMalloc(&p4[b], (size_t)(nkb * npad * 16)));
    CK(hipMalloc(&k1[b], (size_t)(nblk * npad * 16)));
  }
  CK(hipMalloc(&sa, (size_t)n * n * 4));
  CK(hipMalloc(&sb, (size_t)n * n * 4));
  CK(hipMalloc(&flag, 64));
  CK(hipMemset(flag, 0, 64));
  hipLaunchKernelGGL(fill_kernel, dim3(8192), dim3(256), 0, 0, x, v, (int64_t)n, ld, 777u, 0x18000000u);
  CK(hipDeviceSynchronize());

  // S of the shipped path
  CK(launch_pack_fp4(x, 0, ld, v, n, p4[0], flag, 0, nkb));
  CK(hipMemset(sa, 0, (size_t)n * n * 4));
  if (launch_gram_packed_lockstep(p4[0], 1, nkb * 32, n, sa, num_cu, 0) != hipSuccess) {
    (void)hipGetLastError();
    CK(launch_gram_packed(p4[0], 1, nkb * 32, n, sa, num_cu, 0, nullptr));
  }
  CK(launch_pack_kbits(x, 0, ld, v, n, k1[0], flag, 0, nblk));
  const char* vname[5] = {"enc1-split ring3 (shipped)", "enc0 ring4", "enc1 ring3", "enc1-split ring4", "enc1-split ring3 left0"};
  const int nvar = 5;
  for (int var = 0; var < nvar; ++var)
  for (int mode : {0, 2, 4}) {
    g_kbits_variant = var;
    if (var > 0 && mode == 0) continue;
    CK(hipMemset(sb, 0, (size_t)n * n * 4));
    hipError_t e = launch_gram_kbits(k1[0], v, n, sb, num_cu, 0, mode);
    if (e != hipSuccess) { (void)hipGetLastError(); std::printf("full size: k-bits mode %d does not fit\n", mode); continue; }
    CK(hipDeviceSynchronize());
    const unsigned long long d = count_diff(sa, sb, (int64_t)n * n * 4, cnt);
    std::printf("full size n=%d v=%lld: S(k-bits %s, mode %d) vs S(FP4 path): %s (%llu entries differ)\n", n, (long long)v,
                vname[var], mode, d ? "MISMATCH" : "bit-identical", d);
    bad += d != 0;
    if (d) {  // where: by tile, by wave sub-tile (128 x 64), by MFMA tile, first few entries
      std::vector<int32_t> ha((size_t)n * n), hb((size_t)n * n);
      CK(hipMemcpy(ha.data(), sa, (size_t)n * n * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hb.data(), sb, (size_t)n * n * 4, hipMemcpyDeviceToHost));
      int by_tile[10][10] = {}, by_wave[2][4] = {}, by_mi[4] = {}, by_ni[2] = {}, shown = 0, by_r[16] = {}, by_hi[2] = {};
      long long sum_delta = 0;
      for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
          const int32_t a = ha[(size_t)i * n + j], b = hb[(size_t)i * n + j];
          if (a == b) continue;
          by_tile[i / 256][j / 256]++;
          by_wave[(i % 256) / 128][(j % 256) / 64]++;
          by_mi[(i % 128) / 32]++;
          by_ni[(j % 64) / 32]++;
          const int ri = i % 32;  // row inside the MFMA tile: ri = (r & 3) + 8 * (r >> 2) + 4 * hi
          by_hi[(ri >> 2) & 1]++;
          by_r[(ri & 3) + 4 * (ri >> 3)]++;
          sum_delta += (long long)b - a;
          if (shown++ < 12) std::printf("   S[%d][%d]: FP4 %d, k-bits %d (delta %d)\n", i, j, a, b, b - a);
        }
      std::printf("   sum of deltas %lld; by wave (wm, wn):", sum_delta);
      for (int a = 0; a < 2; ++a) for (int b = 0; b < 4; ++b) std::printf(" %d", by_wave[a][b]);
      std::printf("; by mi:");
      for (int a = 0; a < 4; ++a) std::printf(" %d", by_mi[a]);
      std::printf("; by ni: %d %d; by hi: %d %d; by r:", by_ni[0], by_ni[1], by_hi[0], by_hi[1]);
      for (int a = 0; a < 16; ++a) std::printf(" %d", by_r[a]);
      std::printf("\n   by tile (row block, col block):");
      for (int a = 0; a < 10; ++a) for (int b = a; b < 10; ++b) if (by_tile[a][b]) std::printf(" (%d,%d):%d", a, b, by_tile[a][b]);
      std::printf("\n");
    }
  }
  g_kbits_variant = 0;
  int32_t hflag = 0;
  CK(hipMemcpy(&hflag, flag, 4, hipMemcpyDeviceToHost));
  std::printf("flag word after the pre-passes: %d\n", hflag);

  const double mv = (double)v / 1e6;
  auto line = [&](const char* what, float ms) { std::printf("time  %-58s %8.3f ms  (%.3f ms per 10^6 variants)\n", what, ms, ms / mv); };
  // pre-passes
  line("pack_fp4<float> (shipped)", time_ms(0, reps, [&] { CK(launch_pack_fp4(x, 0, ld, v, n, p4[0], flag, 0, nkb)); }));
  line("pack_kbits<float>", time_ms(0, reps, [&] { CK(launch_pack_kbits(x, 0, ld, v, n, k1[0], flag, 0, nblk)); }));
  {
    const int64_t ld8 = (n + 7) / 8 * 8;
    uint8_t* x8;
    CK(hipMalloc(&x8, (size_t)(v * ld8)));
    hipLaunchKernelGGL(to_u8_kernel, dim3(8192), dim3(256), 0, 0, x, x8, v, (int64_t)n, ld, ld8);
    line("pack_u8x8_fp4 (shipped)", time_ms(0, reps, [&] { CK(launch_pack_fp4(x8, 1, ld8, v, n, p4[1], flag, 0, nkb)); }));
    line("pack_u8x8_kbits", time_ms(0, reps, [&] { CK(launch_pack_kbits(x8, 1, ld8, v, n, k1[1], flag, 0, nblk)); }));
    std::printf("       u8 operand == fp32 operand: %s\n", count_diff(k1[0], k1[1], nblk * npad * 16, cnt) ? "MISMATCH" : "ok");
    CK(hipFree(x8));
    const int64_t ldw = ((n + 31) / 32 + 3) / 4 * 4;
    uint32_t* bits;
    CK(hipMalloc(&bits, (size_t)(v * ldw * 4)));
    hipLaunchKernelGGL(to_bits_kernel, dim3(8192), dim3(256), 0, 0, x, bits, v, (int64_t)n, ld, ldw);
    line("expand_bits_fp4 (shipped)", time_ms(0, reps, [&] { CK(launch_expand_bits_fp4(bits, ldw, v, n, p4[1], 0, nkb)); }));
    line("transpose_bits_kbits", time_ms(0, reps, [&] { CK(launch_transpose_bits_kbits(bits, ldw, v, n, k1[1], 0, nblk)); }));
    std::printf("       bitset operand == fp32 operand: %s\n", count_diff(k1[0], k1[1], nblk * npad * 16, cnt) ? "MISMATCH" : "ok");
    CK(hipFree(bits));
  }
  // contractions, whole chip and half chip
  CK(launch_pack_fp4(x, 0, ld, v, n, p4[1], flag, 0, nkb));
  CK(launch_pack_kbits(x, 0, ld, v, n, k1[1], flag, 0, nblk));
  for (int cus : {num_cu, num_cu / 2}) {
    std::string tag = cus == num_cu ? " [whole chip]" : " [sized for half the chip]";
    line(("FP4 lock-step" + tag).c_str(), time_ms(0, reps, [&] { CK(launch_gram_packed_lockstep(p4[0], 1, nkb * 32, n, sa, cus, 0)); }));
    for (int var = 0; var < nvar; ++var) {
      g_kbits_variant = var;
      line((std::string("k-bits ") + vname[var] + " lock-step (mode 2)" + tag).c_str(), time_ms(0, reps, [&] { CK(launch_gram_kbits(k1[0], v, n, sb, cus, 0, 2)); }));
      line((std::string("k-bits ") + vname[var] + " even split (mode 4)" + tag).c_str(), time_ms(0, reps, [&] { CK(launch_gram_kbits(k1[0], v, n, sb, cus, 0, 4)); }));
    }
    g_kbits_variant = 0;
  }
  for (int cus : {192, 160, 144}) {
    std::string tag = " [" + std::to_string(cus) + " workgroups]";
    line(("k-bits even split (mode 4)" + tag).c_str(), time_ms(0, reps, [&] { CK(launch_gram_kbits(k1[0], v, n, sb, cus, 0, 4)); }));
  }
  line("FP4 split-K launch (legacy)", time_ms(0, reps, [&] { CK(launch_gram_packed(p4[0], 1, nkb * 32, n, sa, num_cu, 0, nullptr)); }));
  line("k-bits split-K launch (mode 0)", time_ms(0, reps, [&] { CK(launch_gram_kbits(k1[0], v, n, sb, num_cu, 0, 0)); }));

  // the pipelined step: pre-pass of buffer k+1 (pre-pass stream, behind a 10-us one-wave spin) beside the contraction of buffer k
  hipStream_t ps, gs;
  CK(hipStreamCreateWithFlags(&ps, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&gs, hipStreamNonBlocking));
  hipEvent_t packed[2], consumed[2];
  for (int b = 0; b < 2; ++b) {
    CK(hipEventCreateWithFlags(&packed[b], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&consumed[b], hipEventDisableTiming));
  }
  auto pipeline = [&](const char* what, int fmt, int gram_mode, int gram_cus, int delay_us) {
    const int K = 12;
    double best = 1e30;
    for (int round = 0; round < 3; ++round) {
      CK(hipDeviceSynchronize());
      const double t0 = now_ms();
      for (int k = 0; k < K; ++k) {
        const int b = k & 1;
        if (k >= 2) CK(hipStreamWaitEvent(ps, consumed[b], 0));
        if (k >= 1 && delay_us > 0) hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, ps, (long long)delay_us * 100);
        if (fmt == 1) CK(launch_pack_fp4(x, 0, ld, v, n, p4[b], flag, ps, nkb));
        else CK(launch_pack_kbits(x, 0, ld, v, n, k1[b], flag, ps, nblk));
        CK(hipEventRecord(packed[b], ps));
        CK(hipStreamWaitEvent(gs, packed[b], 0));
        if (fmt == 1) CK(launch_gram_packed_lockstep(p4[b], 1, nkb * 32, n, sa, gram_cus, gs));
        else CK(launch_gram_kbits(k1[b], v, n, sb, gram_cus, gs, gram_mode));
        CK(hipEventRecord(consumed[b], gs));
      }
      CK(hipDeviceSynchronize());
      best = std::min(best, (now_ms() - t0) / K);
    }
    std::printf("pipe  %-58s %8.3f ms per step (%.3f per 10^6 variants, %.0f M variants/s)\n", what, best, best / mv, v / best / 1e3);
  };
  pipeline("FP4: pack_fp4 || lock-step on half the chip (shipped)", 1, 2, num_cu / 2, 10);
  pipeline("k-bits: pack_kbits || lock-step on half the chip", 2, 2, num_cu / 2, 10);
  pipeline("k-bits: pack_kbits || even split, 128 workgroups", 2, 4, 128, 10);
  pipeline("k-bits: pack_kbits || even split, 96 workgroups", 2, 4, 96, 10);
  pipeline("k-bits: pack_kbits || even split, 160 workgroups", 2, 4, 160, 10);
  pipeline("k-bits: pack_kbits || even split, 256 workgroups", 2, 4, num_cu, 10);
  pipeline("k-bits: pack_kbits || lock-step whole chip", 2, 2, num_cu, 10);
  pipeline("k-bits: no head start, lock-step on half the chip", 2, 2, num_cu / 2, 0);
  // serial reference on one stream
  line("serial step FP4 (pack + lock-step, one stream)", time_ms(0, reps, [&] {
         CK(launch_pack_fp4(x, 0, ld, v, n, p4[0], flag, 0, nkb));
         CK(launch_gram_packed_lockstep(p4[0], 1, nkb * 32, n, sa, num_cu, 0));
       }));
  line("serial step k-bits (pack + even split, one stream)", time_ms(0, reps, [&] {
         CK(launch_pack_kbits(x, 0, ld, v, n, k1[0], flag, 0, nblk));
         CK(launch_gram_kbits(k1[0], v, n, sb, num_cu, 0, 4));
       }));
  std::printf("%s\n", bad ? "RESULT: FAILED" : "RESULT: ok");
  return bad ? 1 : 0;
}
