// gram_f32.hip -- S += X^T X on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// Replaces the hot loop of getSimilarityMatrix (reference VariantsPca.scala:184-190):
//     for (c1 <- callset; c2 <- callset) matrix(c1, c2) += 1      per variant, per partition
// With X[v][i] = carrier multiplicity of sample i at variant v (0/1), the sum over variants of the
// ordered-pair indicator is exactly S = X^T X, a dense contraction over the variant axis.
//
// Exactness: X holds small non-negative integers, products and partial sums are integers; an fp32
// accumulator is exact below 2^24, and a launch never feeds more than 2^24 variants through one
// accumulator chain (enforced by the caller).  The epilogue converts to int32 and adds into the
// global int32 partial with integer atomics, so the result is independent of split-K order, of the
// workgroup->XCD placement and of the number of GPUs: bit-identical to the reference's Int matrix.
//
// Shape: N (samples) is small (2,504), V (variants) is huge => a tall-K GEMM with only
// T(T+1)/2 upper-triangular 128x128 output tiles (T = ceil(N/128) = 20 -> 210 tiles).  The chip is
// filled by split-K: grid = tiles x splitk workgroups, each streaming its own variant range.
// The lower triangle is never computed; pcoa_gram_finalize mirrors it once.
//
// Data path per workgroup (256 threads = 4 waves as 2x2, each wave a 64x64 block = 2x2 MFMA tiles):
//   HBM --global_load_lds (16 B/lane, LDS-DMA, no VGPR round trip)--> LDS [BK][128] per panel
//   The LDS image equals the global layout (k-major rows of 128 consecutive samples), which is
//   exactly the operand layout of v_mfma_f32_32x32x2_f32 for BOTH operands of X^T X:
//     A[i][k] = X[k][i0+i]  -> lane l reads  As[k + (l>>5)][i_off + (l&31)]
//     B[k][j] = X[k][j0+j]  -> lane l reads  Bs[k + (l>>5)][j_off + (l&31)]
//   i.e. 32 consecutive dwords per half-wave: conflict-free ds_read_b32, no transpose, no swizzle.
//   Double-buffered: stage s+1 is in flight (counted vmcnt) while stage s feeds the MFMAs.
//
// Roofline: MFMA-bound.  Algorithmic intensity N/2 = 1252 flop/B >> 157.3 TF / 8 TB/s = 20 flop/B.
#include "pcoa_internal.h"

namespace pcoa {
namespace {

constexpr int BM = 128;   // tile edge in samples
constexpr int BK = 16;    // variants per LDS stage
constexpr int NT = 256;   // threads per workgroup

typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__device__ __forceinline__ void wg_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// One LDS stage = panel I ([BK][BM] floats) followed by panel J.
struct Stage {
  float p[2][BK][BM];
};

// Issues the LDS-DMA loads of one stage.  VEC = floats per lane per instruction (4 -> dwordx4).
template <int VEC>
__device__ __forceinline__ void issue_stage(Stage* st, const float* __restrict__ x, int64_t ld, int64_t k0,
                                            int64_t k_end, int col_i, int col_j,
                                            const float* __restrict__ zeros, int wave, int lane) {
  // Diagonal tiles (col_i == col_j) load the same panel twice: 20 of 210 tiles at N = 2504, the
  // second copy is an L2 hit, and the main loop stays branch-free.
#pragma unroll
  for (int pnl = 0; pnl < 2; ++pnl) {
    const int c0 = pnl == 0 ? col_i : col_j;
    if constexpr (VEC == 4) {
      // one instruction moves 2 rows x 128 floats: lanes 0-31 row r, lanes 32-63 row r+1
#pragma unroll
      for (int q = 0; q < BK / 8; ++q) {
        const int pair = wave * (BK / 8) + q;
        const int64_t row = k0 + 2 * pair + (lane >> 5);
        const int64_t col = c0 + (lane & 31) * 4;
        const bool ok = (row < k_end) && (col < ld);
        const float* src = ok ? x + row * ld + col : zeros + lane * 4;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)&st->p[pnl][2 * pair][0], 16, 0, 0);
      }
    } else {
      // one instruction moves 64 floats = half a row
#pragma unroll
      for (int q = 0; q < BK / 2; ++q) {
        const int h = wave * (BK / 2) + q;  // half-row index 0 .. 2*BK-1
        const int64_t row = k0 + (h >> 1);
        const int64_t col = c0 + (h & 1) * 64 + lane;
        const bool ok = (row < k_end) && (col < ld);
        const float* src = ok ? x + row * ld + col : zeros + lane;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)&st->p[pnl][h >> 1][(h & 1) * 64], 4, 0, 0);
      }
    }
  }
}

__device__ __forceinline__ void compute_stage(const Stage* st, int wm, int wn, int lane,
                                              f32x16& c00, f32x16& c01, f32x16& c10, f32x16& c11) {
  const float* as = &st->p[0][0][0];
  const float* bs = &st->p[1][0][0];
  const int l31 = lane & 31;
  const int hi = lane >> 5;
#pragma unroll
  for (int kk = 0; kk < BK; kk += 2) {
    const int r = (kk + hi) * BM;
    const float a0 = as[r + wm * 64 + l31];
    const float a1 = as[r + wm * 64 + 32 + l31];
    const float b0 = bs[r + wn * 64 + l31];
    const float b1 = bs[r + wn * 64 + 32 + l31];
    c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, c00, 0, 0, 0);
    c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, c01, 0, 0, 0);
    c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, c10, 0, 0, 0);
    c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, c11, 0, 0, 0);
  }
}

__device__ __forceinline__ void store_tile(const f32x16& c, int32_t* __restrict__ s32, int n, int row0,
                                           int col0, int lane, bool& inexact) {
  // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  const int j = col0 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int v = (int)c[r];
    // an fp32 chain of integer products is exact below 2^24; beyond it (carrier multiplicities m with V * m^2 >= 2^24 in
    // one launch) the sum may have been rounded: reported, never silent
    inexact |= (j >= i && j < n) && !(__builtin_fabsf(c[r]) < 16777216.0f);   // (padding columns may hold anything)
    if (j >= i && j < n && v != 0) {                                          // upper triangle only
      // (the returned old value costs a round trip, but this kernel has no pre-pass that could bound the carrier
      // multiplicities: an int32 partial that would wrap is reported through the same flag)
      const int old = atomicAdd(&s32[(int64_t)i * n + j], v);
      const int64_t sum = (int64_t)old + (int64_t)v;
      inexact |= (sum > 2147483647LL) || (sum < -2147483648LL);
    }
  }
}

template <int VEC>
__global__ __launch_bounds__(NT) void gram_f32_kernel(const float* __restrict__ x, int64_t ld, int64_t nv,
                                                      int n, int ntile, int ntri, int splitk, int64_t kchunk,
                                                      int32_t* __restrict__ s32, int32_t* __restrict__ flag,
                                                      const float* __restrict__ zeros, int xcd_map) {
  __shared__ __attribute__((aligned(16))) Stage lds[2];

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;

  // blockIdx -> (tile, k-slice).  With xcd_map all tiles of one k-slice run on one XCD (block b is
  // dispatched to XCD b % 8), so that XCD's L2 serves each X row to every tile that needs it.
  int tile, ks;
  const int b = blockIdx.x;
  if (xcd_map) {
    const int q = b >> 3;
    ks = (b & 7) + kNumXcd * (q / ntri);
    tile = q % ntri;
  } else {
    tile = b % ntri;
    ks = b / ntri;
  }
  int ti = 0, rem = tile;
  while (rem >= ntile - ti) {
    rem -= ntile - ti;
    ++ti;
  }
  const int tj = ti + rem;

  const int64_t k_begin = (int64_t)ks * kchunk;
  const int64_t k_end = (k_begin + kchunk < nv) ? (k_begin + kchunk) : nv;
  if (k_begin >= k_end) return;
  // stages are processed in pairs (one per LDS buffer); a trailing odd stage is zero-filled
  const int nstage2 = (int)((k_end - k_begin + 2 * BK - 1) / (2 * BK)) * 2;

  const int col_i = ti * BM, col_j = tj * BM;
  // diagonal tiles: wave (1, 0) covers rows 64..127 x cols 0..63, entirely below the diagonal -> no MFMAs
  const bool idle = (col_i + wm * 64) > (col_j + wn * 64 + 63);

  f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};

  constexpr int LD = 2 * ((VEC == 4) ? (BK / 8) : (BK / 2));  // DMA instructions per wave per stage

  issue_stage<VEC>(&lds[0], x, ld, k_begin, k_end, col_i, col_j, zeros, wave, lane);

  for (int s = 0; s < nstage2; s += 2) {
    // stage s+1 -> buffer 1 goes in flight; stage s (buffer 0) must have landed
    issue_stage<VEC>(&lds[1], x, ld, k_begin + (int64_t)(s + 1) * BK, k_end, col_i, col_j, zeros, wave, lane);
    wait_vmcnt<LD>();
    wg_barrier();
    if (!idle) compute_stage(&lds[0], wm, wn, lane, c00, c01, c10, c11);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wg_barrier();  // every wave is done reading buffer 0
    // stage s+2 -> buffer 0 goes in flight (zero rows past k_end); stage s+1 must have landed
    issue_stage<VEC>(&lds[0], x, ld, k_begin + (int64_t)(s + 2) * BK, k_end, col_i, col_j, zeros, wave, lane);
    wait_vmcnt<LD>();
    wg_barrier();
    if (!idle) compute_stage(&lds[1], wm, wn, lane, c00, c01, c10, c11);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wg_barrier();  // every wave is done reading buffer 1
  }
  wait_vmcnt<0>();  // drain the last (all-zero) prefetch before the LDS is released

  const int row0 = col_i + wm * 64, col0 = col_j + wn * 64;
  bool inexact = false;
  store_tile(c00, s32, n, row0, col0, lane, inexact);
  store_tile(c01, s32, n, row0, col0 + 32, lane, inexact);
  store_tile(c10, s32, n, row0 + 32, col0, lane, inexact);
  store_tile(c11, s32, n, row0 + 32, col0 + 32, lane, inexact);
  if (inexact && flag) atomicOr(flag, 16);
}

}  // namespace

hipError_t launch_gram_f32(const GramLaunch& g, int* splitk_out) {
  if (g.nv <= 0) return hipSuccess;
  const int ntile = (g.n + BM - 1) / BM;
  const int64_t ntri64 = (int64_t)ntile * (ntile + 1) / 2;
  if (ntri64 > (1 << 28)) return hipErrorInvalidValue;
  const int ntri = (int)ntri64;
  const int64_t total_stages = (g.nv + BK - 1) / BK;
  // Aim at ~48 work units per CU (4-5 workgroups are co-resident per CU, so the tail of the last
  // round is a few percent), but keep >= 8 stages per workgroup so the epilogue atomics amortise.
  const int64_t target = (int64_t)(g.num_cu > 0 ? g.num_cu : 256) * 48;
  int64_t splitk = target / ntri;
  const int64_t max_by_work = total_stages / 8;
  if (splitk > max_by_work) splitk = max_by_work;
  if (splitk < 1) splitk = 1;
  int xcd_map = 0;
  if (splitk >= kNumXcd) {
    splitk = (splitk / kNumXcd) * kNumXcd;
    xcd_map = 1;
  }
  const int64_t stages_per = (total_stages + splitk - 1) / splitk;
  const int64_t kchunk = stages_per * BK;
  const int64_t nblocks = (int64_t)ntri * splitk;
  if (nblocks > 0x7fffffffLL) return hipErrorInvalidValue;
  if (splitk_out) *splitk_out = (int)splitk;
  const bool vec4 = ((g.ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.x) & 15) == 0);
  dim3 grid((unsigned)nblocks), block(NT);
  if (vec4) {
    hipLaunchKernelGGL(gram_f32_kernel<4>, grid, block, 0, g.stream, g.x, g.ld, g.nv, g.n, ntile, ntri,
                       (int)splitk, kchunk, g.s32, g.flag, g.zeros, xcd_map);
  } else {
    hipLaunchKernelGGL(gram_f32_kernel<1>, grid, block, 0, g.stream, g.x, g.ld, g.nv, g.n, ntile, ntri,
                       (int)splitk, kchunk, g.s32, g.flag, g.zeros, xcd_map);
  }
  return hipGetLastError();
}

}  // namespace pcoa
