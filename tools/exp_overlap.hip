// exp_overlap.hip -- A/B harness for the fp32 boundary of the Gram path (DESIGN_HISTORY.md 4.1).
//
// Measures, on one MI355X and on the kernels of spark-examples_amd/csrc/gram_packed.hip themselves (the file is
// included, so the anonymous-namespace kernels are visible):
//   1. pack_fp4_kernel (shipped pre-pass) vs pack_fp4_ring_kernel (persistent, LDS-DMA ring, <= 64 VGPRs, 64 KiB LDS):
//      bit-identical operand? time per 10^6 variants at several grid sizes, nt / default cache policy;
//   2. the contraction alone;
//   3. a step = pre-pass + contraction, serial on one stream (what ships) vs pipelined on two streams with ping-pong
//      operand buffers: pre-pass of chunk k+1 beside the contraction of chunk k, for both pre-pass kernels, with and
//      without stream priorities, and on disjoint CU masks.
// Not part of the product; build: make -C tools exp_overlap (hipcc, gfx950).  Prints one line per measurement.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../spark-examples_amd/csrc/gram_packed.hip"

// the library reads its environment knobs through this accessor (pcoa_capi.hip); the harness has none
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

__global__ void fill_kernel(float* x, int64_t count, uint32_t seed, uint32_t thr) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u ^ (uint32_t)(i >> 32) * 40503u ^ seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    x[i] = (h < thr) ? 1.0f : 0.0f;
  }
}

__global__ void diff_kernel(const uint32_t* a, const uint32_t* b, int64_t words, unsigned long long* out) {
  unsigned long long d = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (int64_t)gridDim.x * blockDim.x)
    d += (a[i] != b[i]);
  if (d) atomicAdd(out, d);
}

__global__ void delay_kernel(long long cycles) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
}
// one wave that watches the shader clock: a fixed chain of dependent VALU operations (4 cycles each on a wave64) timed
// with s_memrealtime (100 MHz); s_memtime (clock64) is reported beside it
__global__ void clock_probe_kernel(long long iters, long long* out) {
  const long long w0 = wall_clock64(), c0 = clock64();
  unsigned v = threadIdx.x;
  for (long long i = 0; i < iters; ++i) {
    asm volatile("v_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1\n\t"
                 "v_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1" : "+v"(v));
  }
  if (threadIdx.x == 0) {
    out[0] = clock64() - c0;
    out[1] = wall_clock64() - w0;
    out[2] = v;
  }
}
static int g_delay_us = 0;       // > 0: a one-wave spin of that length in front of every pre-pass (head start for the contraction)
static int g_gram_cus_hint = 0;  // > 0: size the lock-step launch for that many CUs

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Ctx {
  int n = 2504, npad = 0, num_cu = 256;
  int64_t v = 1000000, ld = 2504, nkb = 0;
  float* x = nullptr;
  int8_t* p[2] = {nullptr, nullptr};
  int32_t *s32 = nullptr, *flag = nullptr;
  unsigned long long* cnt = nullptr;
};

enum PackKind { PACK_OLD = 0, PACK_RING = 1 };
struct PackCfg {
  PackKind kind;
  int wgs;  // ring: grid size
  int nt;
  std::string name() const {
    if (kind == PACK_OLD) return "old";
    return "ring" + std::string((nt & 2) ? "8" : "16") + "(wgs=" + std::to_string(wgs) + ((nt & 1) ? ",nt)" : ",dflt)");
  }
};

static void pack(const Ctx& c, const PackCfg& k, int buf, hipStream_t s, int64_t nv = -1) {
  if (nv < 0) nv = c.v;
  if (g_delay_us > 0) hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, s, (long long)g_delay_us * 100);  // wall_clock64: 100 MHz
  if (k.kind == PACK_OLD) CK(launch_pack_fp4(c.x, 0, c.ld, nv, c.n, c.p[buf], c.flag, s, c.nkb));
  else CK(launch_pack_fp4_ring(c.x, c.ld, nv, c.n, c.p[buf], c.flag, s, c.nkb, k.wgs, k.nt));
}
static int g_lockstep = 0;  // 1: lock-step contraction launch (all tiles of 4 k-streams resident, one workgroup per CU)
static void contract(const Ctx& c, int buf, hipStream_t s, int num_cu = 0) {
  if (g_lockstep) {
    if (launch_gram_packed_lockstep(c.p[buf], 1, c.nkb * 32, c.n, c.s32, g_gram_cus_hint ? g_gram_cus_hint : c.num_cu, s) != hipSuccess) {
      (void)hipGetLastError();
      CK(launch_gram_packed(c.p[buf], 1, c.nkb * 32, c.n, c.s32, c.num_cu, s, nullptr));  // shape does not fit: shipped launch
    }
  } else {
    CK(launch_gram_packed(c.p[buf], 1, c.nkb * 32, c.n, c.s32, num_cu ? num_cu : c.num_cu, s, nullptr));
  }
}

static float time_events(hipStream_t s, int reps, const std::function<void()>& body) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  body();  // warm-up
  CK(hipStreamSynchronize(s));
  CK(hipEventRecord(a, s));
  for (int i = 0; i < reps; ++i) body();
  CK(hipEventRecord(b, s));
  CK(hipStreamSynchronize(s));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  CK(hipEventDestroy(a));
  CK(hipEventDestroy(b));
  return ms / reps;
}

// pipelined job: K chunks, pre-pass on sp, contraction on sg, ping-pong operand buffers
static double pipelined(const Ctx& c, const PackCfg& k, hipStream_t sp, hipStream_t sg, int K, int gram_cus = 0) {
  hipEvent_t ev_p[2], ev_g[2];
  for (int b = 0; b < 2; ++b) {
    CK(hipEventCreateWithFlags(&ev_p[b], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ev_g[b], hipEventDisableTiming));
  }
  auto run = [&](int chunks) {
    for (int i = 0; i < chunks; ++i) {
      const int b = i & 1;
      if (i >= 2) CK(hipStreamWaitEvent(sp, ev_g[b], 0));  // the contraction that read this buffer has finished
      pack(c, k, b, sp);
      CK(hipEventRecord(ev_p[b], sp));
      CK(hipStreamWaitEvent(sg, ev_p[b], 0));
      contract(c, b, sg, gram_cus);
      CK(hipEventRecord(ev_g[b], sg));
    }
    CK(hipStreamSynchronize(sp));
    CK(hipStreamSynchronize(sg));
  };
  run(3);
  const double t0 = now_ms();
  run(K);
  const double dt = now_ms() - t0;
  for (int b = 0; b < 2; ++b) {
    CK(hipEventDestroy(ev_p[b]));
    CK(hipEventDestroy(ev_g[b]));
  }
  return dt / K;
}

// the same pipelined job with an event pair around every kernel: prints when each kernel ran (ms since the first one)
static double timeline(const Ctx& c, const PackCfg& k, hipStream_t sp, hipStream_t sg, int K, bool verbose = true) {
  std::vector<hipEvent_t> pa(K), pb(K), ga(K), gb(K);
  for (int i = 0; i < K; ++i) {
    CK(hipEventCreate(&pa[i])); CK(hipEventCreate(&pb[i])); CK(hipEventCreate(&ga[i])); CK(hipEventCreate(&gb[i]));
  }
  for (int i = 0; i < K; ++i) {
    const int b = i & 1;
    if (i >= 2) CK(hipStreamWaitEvent(sp, gb[i - 2], 0));
    CK(hipEventRecord(pa[i], sp));
    pack(c, k, b, sp);
    CK(hipEventRecord(pb[i], sp));
    CK(hipStreamWaitEvent(sg, pb[i], 0));
    CK(hipEventRecord(ga[i], sg));
    contract(c, b, sg);
    CK(hipEventRecord(gb[i], sg));
  }
  CK(hipStreamSynchronize(sp));
  CK(hipStreamSynchronize(sg));
  float span = 0, psum = 0, gsum = 0;
  CK(hipEventElapsedTime(&span, pa[1], pa[K - 1]));
  for (int i = 1; i < K - 1; ++i) {
    float t;
    CK(hipEventElapsedTime(&t, pa[i], pb[i])); psum += t;
    CK(hipEventElapsedTime(&t, ga[i], gb[i])); gsum += t;
  }
  const double step = span / (K - 2);
  std::printf("pipelined %-24s + %-9s contraction: steady state %.3f ms/step (pre-pass kernel %.3f ms, contraction %.3f ms from queue entry)\n",
              k.name().c_str(), g_lockstep ? "lock-step" : "shipped", step, psum / (K - 2), gsum / (K - 2));
  if (verbose) {
    for (int i = 0; i < K; ++i) {
      float a0, a1, b0, b1;
      CK(hipEventElapsedTime(&a0, pa[0], pa[i])); CK(hipEventElapsedTime(&a1, pa[0], pb[i]));
      CK(hipEventElapsedTime(&b0, pa[0], ga[i])); CK(hipEventElapsedTime(&b1, pa[0], gb[i]));
      std::printf("  %2d  pack %7.3f..%7.3f (%.3f) | gram %7.3f..%7.3f (%.3f)\n", i, a0, a1, a1 - a0, b0, b1, b1 - b0);
    }
  }
  for (int i = 0; i < K; ++i) {
    CK(hipEventDestroy(pa[i])); CK(hipEventDestroy(pb[i])); CK(hipEventDestroy(ga[i])); CK(hipEventDestroy(gb[i]));
  }
  return step;
}

int main(int argc, char** argv) {
  Ctx c;
  if (argc > 1) c.v = std::atoll(argv[1]);
  const int K = argc > 2 ? std::atoi(argv[2]) : 12;
  CK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  c.num_cu = prop.multiProcessorCount;
  c.npad = (int)gram_packed_npad(c.n);
  c.nkb = gram_kb_pad(c.v, 1);
  std::printf("device %s, %d CUs; N = %d, V = %lld per chunk, %lld k-blocks\n", prop.gcnArchName, c.num_cu, c.n,
              (long long)c.v, (long long)c.nkb);
  const size_t pbytes = (size_t)(c.nkb + 24) * c.npad * 16;
  CK(hipMalloc((void**)&c.x, sizeof(float) * (size_t)c.v * c.ld + 4096));
  for (int b = 0; b < 2; ++b) CK(hipMalloc((void**)&c.p[b], pbytes));
  CK(hipMalloc((void**)&c.s32, sizeof(int32_t) * (size_t)c.n * c.n));
  CK(hipMalloc((void**)&c.flag, 64));
  CK(hipMalloc((void**)&c.cnt, 64));
  CK(hipMemset(c.s32, 0, sizeof(int32_t) * (size_t)c.n * c.n));
  CK(hipMemset(c.flag, 0, 64));
  hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, c.x, (int64_t)c.v * c.ld, 12345u, 0x1C000000u);  // ~11 % carriers
  CK(hipDeviceSynchronize());

  hipStream_t s0;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));

  if (argc > 3 && std::string(argv[3]) == "clocks") {
    // ---- what the shader clock does under each load: a one-wave probe on its own stream beside (a) nothing, (b) the
    // pre-pass, (c) the lock-step contraction on half the chip, (d) both (the pipeline's steady state)
    hipStream_t sp, sg, sc;
    CK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sg, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sc, hipStreamNonBlocking));
    long long* probe = nullptr;
    CK(hipMalloc((void**)&probe, 64));
    const PackCfg old_pack{PACK_OLD, 0, 0};
    pack(c, old_pack, 0, s0);
    pack(c, old_pack, 1, s0);
    CK(hipStreamSynchronize(s0));
    g_lockstep = 1;
    g_gram_cus_hint = 128;
    const char* names[] = {"idle", "pre-pass alone", "contraction alone (112 CUs)", "pre-pass + contraction", "contraction alone (whole chip)"};
    for (int round = 0; round < 2; ++round)
      for (int mode = 0; mode < 5; ++mode) {
        g_gram_cus_hint = (mode == 4) ? 0 : 128;
        const int reps = 6;
        // load first (a few launches deep), then the probe for 4 ms in the middle of it
        if (mode == 1 || mode == 3) for (int i = 0; i < reps; ++i) pack(c, old_pack, 0, sp);
        if (mode == 2 || mode == 3) for (int i = 0; i < reps; ++i) contract(c, 1, sg);
        if (mode == 4) for (int i = 0; i < 2 * reps; ++i) contract(c, 1, sg);
        hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, sc, 150000LL);        // 1.5 ms: let the load ramp up
        const long long iters = 250000;   // 2 M dependent adds = 8 M cycles: ~3.5 ms at 2.4 GHz
        hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, sc, iters, probe);
        long long h[3] = {0, 0, 0};
        CK(hipMemcpyAsync(h, probe, 24, hipMemcpyDeviceToHost, sc));
        CK(hipStreamSynchronize(sc));
        CK(hipStreamSynchronize(sp));
        CK(hipStreamSynchronize(sg));
        std::printf("[clock] %-32s 2,000,000 dependent v_add_u32 in %.3f ms -> %.0f MHz at 4 cycles each; s_memtime / s_memrealtime = %.3f\n",
                    names[mode], (double)h[1] * 1e-5, 8.0 * (double)iters * 4.0 / ((double)h[1] * 1e-2), (double)h[0] / (double)h[1]);
        std::fflush(stdout);
      }
    std::printf("done\n");
    return 0;
  }
  if (argc > 3 && std::string(argv[3]) == "balanced") {
    // ---- focused A/B: tile deal of the half-chip lock-step contraction (row-major cut vs 6 panels per XCD), alone and in
    // the shipped pipeline (no masks, 10-us head start), interleaved rounds
    hipStream_t sp, sg;
    CK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sg, hipStreamNonBlocking));
    const PackCfg old_pack{PACK_OLD, 0, 0};
    pack(c, old_pack, 0, s0);
    pack(c, old_pack, 1, s0);
    g_lockstep = 1;
    g_gram_cus_hint = 128;
    const size_t nn = (size_t)c.n * c.n;
    std::vector<int32_t> a(nn), b(nn);
    for (int map : {2, 3}) {
      g_lockstep_map = map;
      CK(hipMemsetAsync(c.s32, 0, nn * 4, s0));
      contract(c, 0, s0);
      CK(hipMemcpyAsync(map == 2 ? a.data() : b.data(), c.s32, nn * 4, hipMemcpyDeviceToHost, s0));
      CK(hipStreamSynchronize(s0));
    }
    size_t diff = 0; long long sum = 0;
    for (size_t i = 0; i < nn; ++i) { diff += a[i] != b[i]; sum += a[i]; }
    std::printf("balanced deal vs row-major cut: %zu differing entries of %zu (sum %lld)  %s\n", diff, nn, sum,
                diff == 0 && sum > 0 ? "OK" : "MISMATCH");
    for (int round = 0; round < 3; ++round)
      for (int map : {2, 3}) {
        g_lockstep_map = map;
        g_delay_us = 0;
        const float alone = time_events(s0, 4, [&] { contract(c, 0, s0); });
        g_delay_us = 10;
        std::printf("[deal %s] contraction alone (sized for 128 CUs) %.3f ms | ", map == 3 ? "balanced " : "row-major", alone);
        timeline(c, old_pack, sp, sg, K, false);
        std::fflush(stdout);
      }
    std::printf("done\n");
    return 0;
  }

  // ---- 1. bit-identity of the two pre-passes (full tile, and a ragged variant count: rows beyond nv in the last k-block)
  for (int64_t nv : {c.v, c.v - 13}) {
    CK(hipMemsetAsync(c.p[0], 0xAB, pbytes, s0));
    CK(hipMemsetAsync(c.p[1], 0xCD, pbytes, s0));
    pack(c, PackCfg{PACK_OLD, 0, 0}, 0, s0, nv);
    pack(c, PackCfg{PACK_RING, c.num_cu, 1}, 1, s0, nv);
    CK(hipMemsetAsync(c.cnt, 0, 8, s0));
    hipLaunchKernelGGL(diff_kernel, dim3(2048), dim3(256), 0, s0, (const uint32_t*)c.p[0], (const uint32_t*)c.p[1],
                       (int64_t)c.nkb * c.npad * 4, c.cnt);
    unsigned long long d = 0;
    int32_t fl = 0;
    CK(hipMemcpyAsync(&d, c.cnt, 8, hipMemcpyDeviceToHost, s0));
    CK(hipMemcpyAsync(&fl, c.flag, 4, hipMemcpyDeviceToHost, s0));
    CK(hipStreamSynchronize(s0));
    std::printf("identity nv=%lld: %llu differing words of %lld, flag=%d  %s\n", (long long)nv, d,
                (long long)c.nkb * c.npad * 4, fl, (d == 0 && fl == 0) ? "OK" : "MISMATCH");
  }
  {  // a non-binary value must raise flag bit 3 in both kernels
    const float two = 2.0f;
    CK(hipMemcpy(c.x + (size_t)77777 * c.ld + 1234, &two, 4, hipMemcpyHostToDevice));
    for (int kind = 0; kind < 2; ++kind) {
      CK(hipMemsetAsync(c.flag, 0, 64, s0));
      pack(c, kind ? PackCfg{PACK_RING, c.num_cu, 1} : PackCfg{PACK_OLD, 0, 0}, 0, s0);
      int32_t fl = 0;
      CK(hipMemcpyAsync(&fl, c.flag, 4, hipMemcpyDeviceToHost, s0));
      CK(hipStreamSynchronize(s0));
      std::printf("non-binary value, %s: flag=%d %s\n", kind ? "ring" : "old", fl, fl == 8 ? "OK" : "MISSED");
    }
    const float one = 1.0f;
    CK(hipMemcpy(c.x + (size_t)77777 * c.ld + 1234, &one, 4, hipMemcpyHostToDevice));
    CK(hipMemset(c.flag, 0, 64));
  }

  // ---- 2. kernels alone
  const double gb = (4.0 * c.v * c.n + (double)c.nkb * c.npad * 16) / 1e9;
  std::vector<PackCfg> packs = {{PACK_OLD, 0, 0}};
  for (int nt : {1, 0})
    for (int mult : {1, 2, 3, 4}) packs.push_back({PACK_RING, c.num_cu * mult, nt});
  for (const auto& k : packs) {
    const float ms = time_events(s0, 6, [&] { pack(c, k, 0, s0); });
    std::printf("alone  pack %-22s %.3f ms  %.0f GB/s (read+write, algorithmic)\n", k.name().c_str(), ms, gb / ms * 1e3);
  }
  pack(c, packs[0], 0, s0);
  pack(c, packs[0], 1, s0);
  int least = 0, greatest = 0;
  CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
  hipStream_t sp, sg, sp_lo, sg_hi;
  CK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&sg, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&sp_lo, hipStreamNonBlocking, least));
  CK(hipStreamCreateWithPriority(&sg_hi, hipStreamNonBlocking, greatest));
  // the lock-step launch must give the same S as the shipped one
  {
    const size_t nn = (size_t)c.n * c.n;
    std::vector<int32_t> a(nn), b(nn);
    CK(hipMemsetAsync(c.s32, 0, nn * 4, s0));
    g_lockstep = 0; contract(c, 0, s0);
    CK(hipMemcpyAsync(a.data(), c.s32, nn * 4, hipMemcpyDeviceToHost, s0));
    CK(hipMemsetAsync(c.s32, 0, nn * 4, s0));
    g_lockstep = 1; contract(c, 0, s0);
    CK(hipMemcpyAsync(b.data(), c.s32, nn * 4, hipMemcpyDeviceToHost, s0));
    CK(hipStreamSynchronize(s0));
    size_t diff = 0; long long sum = 0;
    for (size_t i = 0; i < nn; ++i) { diff += a[i] != b[i]; sum += a[i]; }
    std::printf("lock-step contraction vs shipped: %zu differing entries of %zu (sum %lld)  %s\n", diff, nn, sum,
                diff == 0 && sum > 0 ? "OK" : "MISMATCH");
  }
  std::printf("LDS per CU reported: sharedMemPerMultiprocessor %zu, maxSharedMemoryPerMultiProcessor %zu, sharedMemPerBlock %zu\n",
              (size_t)prop.sharedMemPerMultiprocessor, (size_t)prop.maxSharedMemoryPerMultiProcessor, (size_t)prop.sharedMemPerBlock);
  for (int ls : {0, 1}) {
    g_lockstep = ls;
    const char* gname = ls ? "lock-step(220 WGs, split-K 4)" : "shipped(1760 WGs, split-K 32)";
    const float gram_ms = time_events(s0, 6, [&] { contract(c, 0, s0); });
    std::printf("alone  contraction %-32s %.3f ms per launch of %lld variants\n", gname, gram_ms, (long long)c.v);
    const float ms = time_events(s0, K, [&] { pack(c, packs[0], 0, s0); contract(c, 0, s0); });
    std::printf("serial step    [%s], pack old  %.3f ms/step\n", gname, ms);
  }
  // ---- no masks: the contraction (lock-step, 112 workgroups = one per CU on 110 CUs) gets a head start of a few us, the
  // pre-pass then fills every CU the contraction left (its 96-VGPR waves do not fit beside 2 x 224 on a SIMD)
  g_lockstep = 1;
  for (int hint : {128, 256}) {
    g_gram_cus_hint = hint;
    for (int dl : {0, 5, 20, 50, 200}) {
      g_delay_us = dl;
      std::printf("[no masks, contraction sized for %d CUs, pre-pass delayed %3d us] ", hint, dl);
      timeline(c, packs[0], sp, sg, K, false);
      std::fflush(stdout);
    }
  }
  g_delay_us = 20; g_gram_cus_hint = 128;
  timeline(c, packs[0], sp, sg, 6, true);
  g_delay_us = 0; g_gram_cus_hint = 0; g_lockstep = 0;
  // ---- disjoint CU sets.  Mask bit g = (XCD g % 8, CU g / 8 of that XCD) -- decoded from r02a (12 bits per word gave
  // 16 CUs on XCDs 0-3 and 8 on XCDs 4-7).  Pre-pass on CUs [0, pc) of every XCD, contraction on CUs [pc, 32).
  // (pc, gc): pre-pass on CUs [0, pc) of every XCD, contraction on CUs [32 - gc, 32); pc + gc > 32 = overlapping sets
  const int splits[][2] = {{16, 16}};
  for (const auto& sp2 : splits) {
    const int pc = sp2[0], gc = sp2[1];
    uint32_t mp[8] = {0}, mg[8] = {0};
    for (int g = 0; g < 256; ++g) {
      if (g / 8 < pc) mp[g / 32] |= 1u << (g % 32);
      if (g / 8 >= 32 - gc) mg[g / 32] |= 1u << (g % 32);
    }
    hipStream_t smp, smg;
    if (hipExtStreamCreateWithCUMask(&smp, 8, mp) != hipSuccess || hipExtStreamCreateWithCUMask(&smg, 8, mg) != hipSuccess) {
      std::printf("CU-mask streams unavailable\n");
      (void)hipGetLastError();
      break;
    }
    const int gcus = 8 * gc;
    const int saved = c.num_cu;
    for (int ls : {1}) {
      g_lockstep = ls;
      c.num_cu = gcus;  // the contraction sizes its grid for the CUs it owns
      const float galone = time_events(smg, 4, [&] { contract(c, 0, smg); });
      std::printf("[%2d + %2d CUs per XCD] contraction %-9s alone on %3d CUs: %.3f ms\n", pc, gc, ls ? "lock-step" : "shipped",
                  gcus, galone);
      std::vector<PackCfg> cand = {{PACK_OLD, 0, 0}, {PACK_RING, 16 * pc, 0}};
      for (const auto& k : cand) {
        c.num_cu = saved;
        const float alone = time_events(smp, 3, [&] { pack(c, k, 0, smp); });
        c.num_cu = gcus;
        std::printf("   pre-pass alone on %3d CUs %.3f ms (%.1f GB/s per CU) | ", 8 * pc, alone, gb / alone * 1e3 / (8 * pc));
        timeline(c, k, smp, smg, K, false);
        std::fflush(stdout);
      }
    }
    c.num_cu = saved;
    g_lockstep = 0;
    CK(hipStreamDestroy(smp));
    CK(hipStreamDestroy(smg));
  }
  // the same job with the contraction FIRST in every pair does not exist (it depends on the pre-pass); what can differ
  // is which kernel reaches an empty chip first: repeat the best candidates with 2-chunk look-ahead disabled
  // ---- disjoint CU masks: the pre-pass on m CUs per 32, the contraction on the rest
  if (c.num_cu == 256 && argc > 3) {  // third argument present: also the CU-mask runs (r02a: no gain at any split)
    for (int m : {8, 12, 16}) {
      uint32_t mp[8], mg[8];
      for (int j = 0; j < 8; ++j) {
        mp[j] = (m >= 32) ? 0xffffffffu : ((1u << m) - 1u);
        mg[j] = ~mp[j];
      }
      hipStream_t smp, smg;
      if (hipExtStreamCreateWithCUMask(&smp, 8, mp) != hipSuccess || hipExtStreamCreateWithCUMask(&smg, 8, mg) != hipSuccess) {
        std::printf("CU-mask streams unavailable\n");
        (void)hipGetLastError();
        break;
      }
      for (const auto& k : {packs[0], PackCfg{PACK_RING, 8 * m, 1}, PackCfg{PACK_RING, 16 * m, 1}, PackCfg{PACK_RING, 32 * m, 1}}) {
        const float alone = time_events(smp, 3, [&] { pack(c, k, 0, smp); });
        const double step = pipelined(c, k, smp, smg, K, 256 - 8 * m);
        std::printf("CU masks %3d + %3d CUs, pack %-22s alone %.3f ms (%.0f GB/s, %.1f GB/s per CU)   pipelined %.3f ms/step\n",
                    8 * m, 256 - 8 * m, k.name().c_str(), alone, gb / alone * 1e3, gb / alone * 1e3 / (8 * m), step);
        std::fflush(stdout);
      }
      CK(hipStreamDestroy(smp));
      CK(hipStreamDestroy(smg));
    }
  }
  std::printf("done\n");
  return 0;
}
