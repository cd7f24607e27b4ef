// gram_packed.hip -- S += X^T X on the low-precision matrix cores of gfx950, exact.
//
// Same contraction as gram_f32.hip (reference VariantsPca.scala:184-190), computed in the reference's own
// arithmetic: integer counts (`DenseMatrix.zeros[Int]`, :185).  Genotype indicators are 0/1, so nothing is lost in
// low-precision operands as long as the accumulation is exact:
//
//   FMT 1 (default, binary tiles)   MX-FP4 E2M1 operands (0 -> 0x0, 1.0 -> 0x2), v_mfma_f32_32x32x64_f8f6f4 in
//                                   its unscaled form, fp32 accumulators (exact below 2^24; a launch feeds at most
//                                   2^20 variants through one chain), ~10 PFLOP/s dense
//   FMT 0 (carrier multiplicities)  int8 operands (0..127), v_mfma_i32_32x32x32_i8, int32 accumulators, ~5 POP/s
//
// Operand layout P[kb][Npad][16 B], "k-blocked": the 16 bytes at P[kb][i] are sample i's indicators for the 32 (FP4)
// or 16 (int8) variants of k-block kb.  That is exactly one lane's operand slice of the MFMA, for the A operand
// (row i) and for the B operand (column j) alike -- X^T X uses the same k-slot mapping on both sides, so any
// consistent order of the k's inside a block gives the same sum -- and the LDS image IS the global image.
//
// Pre-passes (HBM-bound, one per boundary of include/pcoa.h; all of them zero the padding):
//   pack_fp4_kernel<float|uint8>   dense tile -> FP4; verifies that every value is exactly 0 or 1 (flag bit 3)
//   pack_u8x8_fp4_kernel           uint8 tile, 8-byte loads, byte-gather + spread8
//   expand_bits_fp4_kernel         carrier bitsets (1 bit per genotype) -> FP4 via v_readlane + lane-mask select
//   pack_f32_i8_kernel / pack_u8_i8_kernel / densify_csr_i8_kernel    -> int8 (values 0..127, flag bit 2 otherwise)
//
// gram_packed_kernel<FMT, ...>      P -> S32: upper-triangular 256x256 tiles x split-K, integer atomics.
//   512 threads = 8 waves as 2(M) x 4(N), each wave a 128x64 block = 4x2 MFMA tiles (128 accumulators per lane).
//   One stage = 4 k-blocks x 2 panels x 256 samples x 16 B = 32 KiB, brought in by 32 global_load_lds_dwordx4 (1 KiB
//   each), 3-stage LDS ring (96 KiB), counted vmcnt (two stages stay in flight), raw s_barrier (never
//   __syncthreads, which would drain the DMA queue).  Operand reads are ds_read_b128 of 32 consecutive 16-B slots per
//   half-wave: conflict-free.  Default schedule: ping-pong (two wave groups half a stage apart, see below);
//   PCOA_GRAM_I8_CFG=43 selects the in-phase ring, 143 the
//   ping-pong schedule without the two MFMAs issued behind the phase barrier.
//
// Measured at N = 2504 per 10^6 variants: FP4 1.13-1.16 ms (6 PFLOP/s issued), int8 2.14 ms; DESIGN_HISTORY.md 4.1 / 4.2.
#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "pcoa_internal.h"

namespace pcoa {
namespace {

constexpr int TM = 256;        // padding granule of the packed operand (samples)
constexpr int KB = 16;         // variants per k-block (one lane's operand slice)

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wg_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// ---------------------------------------------------------------------------------------------- pack
// One thread: 16 variants x 4 samples.  A wave covers 256 consecutive samples of one k-block, so
// every load instruction reads 1 KiB contiguous and the wave writes 4 KiB contiguous.
// flag bit 2 is raised for a value that is not an integer in [0, 127].
template <int VEC>
__global__ __launch_bounds__(256) void pack_f32_i8_kernel(const float* __restrict__ x, int64_t ld, int64_t nv,
                                                          int n, int npad, int64_t nkb_pad,
                                                          int8_t* __restrict__ p, int32_t* __restrict__ flag) {
  const int groups = npad >> 2;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t kb = gid / groups;
  const int g = (int)(gid - kb * groups);
  if (kb >= nkb_pad) return;
  const int i0 = g * 4;
  float v[16][4];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int64_t row = kb * KB + t;
    if (row < nv) {
      const float* src = x + row * ld + i0;
      if (VEC == 4 && i0 + 3 < ld) {
        const float4 f = *reinterpret_cast<const float4*>(src);
        v[t][0] = f.x; v[t][1] = f.y; v[t][2] = f.z; v[t][3] = f.w;
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) v[t][s] = (i0 + s < ld) ? src[s] : 0.0f;
      }
    } else {
#pragma unroll
      for (int s = 0; s < 4; ++s) v[t][s] = 0.0f;
    }
  }
  bool bad = false;
  uint32_t w[4][4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const bool live = (i0 + s) < n;  // padding columns [n, ld) may hold anything: forced to zero
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint32_t word = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const float f = live ? v[q * 4 + b][s] : 0.0f;
        const int iv = (int)f;
        bad |= !((float)iv == f && iv >= 0 && iv <= 127);
        word |= ((uint32_t)iv & 0xffu) << (8 * b);
      }
      w[s][q] = word;
    }
  }
  uint4* dst = reinterpret_cast<uint4*>(p + ((size_t)kb * npad + i0) * KB);
  uint32_t mx = 0;  // largest multiplicity of this thread's 64 values (bytewise max of the packed words)
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    dst[s] = make_uint4(w[s][0], w[s][1], w[s][2], w[s][3]);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int b = 0; b < 4; ++b) mx = max(mx, (w[s][q] >> (8 * b)) & 0xffu);
  }
  if (bad) atomicOr(flag, 4);
  if (mx > 1) atomicMax(flag + 1, (int32_t)mx);   // binary tiles (the common case) never touch the word
}

// uint8 twin of the pre-pass: X u8 [V][ld] -> P.  One thread = 16 variants x 4 samples (16 coalesced 4-B
// loads, a 16x4 byte transpose with v_perm, 4 x 16-B stores).  2.5 + 2.56 GB per 10^6 variants.
__global__ __launch_bounds__(256) void pack_u8_i8_kernel(const uint8_t* __restrict__ x, int64_t ld, int64_t nv,
                                                         int n, int npad, int64_t nkb_pad, int8_t* __restrict__ p,
                                                         int32_t* __restrict__ flag, int vec_ok) {
  const int groups = npad >> 2;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t kb = gid / groups;
  const int g = (int)(gid - kb * groups);
  if (kb >= nkb_pad) return;
  const int i0 = g * 4;
  uint32_t v[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int64_t row = kb * KB + t;
    uint32_t w = 0;
    if (row < nv) {
      const uint8_t* src = x + row * ld + i0;
      if (vec_ok && i0 + 3 < ld) {
        w = *reinterpret_cast<const uint32_t*>(src);
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s)
          if (i0 + s < ld) w |= (uint32_t)src[s] << (8 * s);
      }
    }
    // padding columns [n, ld) may hold anything: forced to zero
#pragma unroll
    for (int s = 0; s < 4; ++s)
      if (i0 + s >= n) w &= ~(0xffu << (8 * s));
    v[t] = w;
  }
  uint32_t any = 0, mx = 0;
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    any |= v[t];
#pragma unroll
    for (int b = 0; b < 4; ++b) mx = max(mx, (v[t] >> (8 * b)) & 0xffu);
  }
  if (mx > 1) atomicMax(flag + 1, (int32_t)min(mx, 127u));
  uint4* dst = reinterpret_cast<uint4*>(p + ((size_t)kb * npad + i0) * KB);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    uint32_t o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      o[q] = ((v[4 * q] >> (8 * s)) & 0xffu) | (((v[4 * q + 1] >> (8 * s)) & 0xffu) << 8) |
             (((v[4 * q + 2] >> (8 * s)) & 0xffu) << 16) | (((v[4 * q + 3] >> (8 * s)) & 0xffu) << 24);
    dst[s] = make_uint4(o[0], o[1], o[2], o[3]);
  }
  if (any & 0x80808080u) atomicOr(flag, 4);  // a value above 127
}

// CSR carrier lists (RDD[Seq[Int]], VariantsPca.scala:153-168) straight into the k-blocked operand:
// one wave per variant row, +1 into byte (v % 16) of P[v / 16][sample] through a 32-bit atomic on
// the enclosing word (repeated indices count with multiplicity, as the reference's double loop does;
// a byte that would pass 127 raises flag bit 2).  P must be zero-filled beforehand.
__global__ __launch_bounds__(256) void densify_csr_i8_kernel(const int32_t* __restrict__ idx,
                                                             const int64_t* __restrict__ offs, int64_t nv,
                                                             int64_t offs_base, int8_t* __restrict__ p, int npad,
                                                             int32_t n, int32_t* __restrict__ flag) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nv) return;
  const int64_t b = offs[row] - offs_base, e = offs[row + 1] - offs_base;
  const int64_t kb = row / KB;
  const int t = (int)(row % KB);
  for (int64_t q = b + lane; q < e; q += 64) {
    const int32_t c = idx[q];
    if (c < 0 || c >= n) {
      atomicOr(flag, 1);
      continue;
    }
    uint32_t* word = reinterpret_cast<uint32_t*>(p + ((size_t)kb * npad + c) * KB) + (t >> 2);
    const uint32_t old = atomicAdd(word, 1u << (8 * (t & 3)));
    if (((old >> (8 * (t & 3))) & 0xffu) >= 127u) atomicOr(flag, 4);
  }
}

// CSR carrier lists WITHOUT repeats (the host checks) straight into the FP4 operand: one wave per variant row, the
// nibble of variant (row % 32) in sample c's 16-byte slot of k-block row / 32 is set to 0x2 through a 32-bit atomic
// OR (the 32 rows of a k-block share the slots).  The region must be zero-filled beforehand.
__global__ __launch_bounds__(256) void densify_csr_fp4_kernel(const int32_t* __restrict__ idx,
                                                              const int64_t* __restrict__ offs, int64_t nv,
                                                              int64_t offs_base, int8_t* __restrict__ p, int npad,
                                                              int32_t n, int32_t* __restrict__ flag) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nv) return;
  const int64_t b = offs[row] - offs_base, e = offs[row + 1] - offs_base;
  const int64_t kb = row / 32;
  const int t = (int)(row % 32);
  for (int64_t q = b + lane; q < e; q += 64) {
    const int32_t c = idx[q];
    if (c < 0 || c >= n) {
      atomicOr(flag, 1);
      flag[2] = c;  // one of the offending indices, for the error message (every flag buffer has >= 4 words)
      continue;
    }
    uint32_t* word = reinterpret_cast<uint32_t*>(p + ((size_t)kb * npad + c) * 16) + (t >> 3);
    const uint32_t bit = 2u << (4 * (t & 7));
    if (atomicOr(word, bit) & bit) atomicOr(flag, 32);  // a repeated callset: flag bit 5 (densify_csr_kbits_kernel)
  }
}

// ---- FP4 pre-pass: X (fp32 or uint8, values exactly 0 / 1) -> P4 [V/32][Npad][16 B], 32 nibbles per lane slice.
// One thread: 32 variants x 4 samples, in two halves of 16 variants (8 bytes of each sample's slice per half).
// flag bit 3 (value 8) is raised for a value that is not exactly 0 or 1.
typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <typename T, int VEC, bool NT = false>
__global__ __launch_bounds__(256) void pack_fp4_kernel(const T* __restrict__ x, int64_t ld, int64_t nv, int n, int npad,
                                                       int64_t nkb_pad, int8_t* __restrict__ p,
                                                       int32_t* __restrict__ flag) {
  // a wave = 64 consecutive 4-sample groups of ONE k-block (npad / 4 is a multiple of 64): the k-block index is
  // wave-uniform, which keeps the row addresses in SGPRs (scalar base + one per-lane column offset)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = npad >> 8;  // waves per k-block
  const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
  const int64_t kb = wid / gw;
  const int g = (int)(wid - kb * gw) * 64 + lane;
  if (kb >= nkb_pad) return;
  const int i0 = g * 4;
  bool bad = false;
  uint32_t badw = 0;
  uint32_t w[4][4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int q = 0; q < 4; ++q) w[s][q] = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    uint32_t one[16];  // bit s of one[t] = sample i0+s carries at variant 32*kb + 16*h + t
    if constexpr (VEC == 4) {
      // VEC == 4 means ld % 4 == 0: a group of 4 columns is wholly inside the row or wholly padding.  Branch-free
      // (address clamped into the tile, result masked) and in two steps -- all 16 row loads of the half first, then
      // the arithmetic -- so that 16 loads per lane are in flight; left to itself the compiler waits for every row
      // before loading the next one.
      typedef typename std::conditional<sizeof(T) == 4, f32x4_t, uint32_t>::type Raw;
      Raw raw[16];
      const int64_t col = (i0 < ld) ? i0 : 0;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int64_t row = kb * 32 + h * 16 + t;
        const T* src = x + (row < nv ? row : nv - 1) * ld + col;
        if constexpr (NT) raw[t] = __builtin_nontemporal_load(reinterpret_cast<const Raw*>(src));
        else raw[t] = *reinterpret_cast<const Raw*>(src);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int64_t row = kb * 32 + h * 16 + t;
        const uint32_t valid = (uint32_t)(row < nv) & (uint32_t)(i0 < ld);
        uint32_t bits = 0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {   // integer logic only: `&&` / `||` would come back as branches
          uint32_t is1, is0;
          if constexpr (sizeof(T) == 4) {
            is1 = (uint32_t)(raw[t][s] == 1.0f);
            is0 = (uint32_t)(raw[t][s] == 0.0f);
          } else {
            const uint32_t b = (raw[t] >> (8 * s)) & 0xffu;
            is1 = (uint32_t)(b == 1u);
            is0 = (uint32_t)(b == 0u);
          }
          const uint32_t live = valid & (uint32_t)(i0 + s < n);  // columns [n, ld) may hold anything
          badw |= live & ((is1 | is0) ^ 1u);
          bits |= (live & is1) << s;
        }
        one[t] = bits;
      }
    } else {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int64_t row = kb * 32 + h * 16 + t;
        uint32_t bits = 0;
        if (row < nv) {
          const T* src = x + row * ld + i0;
          T v[4];
#pragma unroll
          for (int s = 0; s < 4; ++s) v[s] = (i0 + s < ld) ? src[s] : (T)0;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            if (i0 + s < n) {  // padding columns [n, ld) may hold anything: ignored
              const bool is1 = (v[s] == (T)1);
              bad |= !(is1 || v[s] == (T)0);
              bits |= (is1 ? 1u : 0u) << s;
            }
          }
        }
        one[t] = bits;
      }
    }
    // nibble of variant t = 0x2 (E2M1 1.0) or 0x0; 8 variants per 32-bit word
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        uint32_t word = 0;
#pragma unroll
        for (int b = 0; b < 8; ++b) word |= (((one[q * 8 + b] >> s) & 1u) << 1) << (4 * b);
        w[s][h * 2 + q] = word;
      }
  }
  uint4* dst = reinterpret_cast<uint4*>(p + ((size_t)kb * npad + i0) * 16);
#pragma unroll
  for (int s = 0; s < 4; ++s) dst[s] = make_uint4(w[s][0], w[s][1], w[s][2], w[s][3]);
  if (bad || badw) atomicOr(flag, 8);
}

#ifdef PCOA_EXPERIMENTS
// ---- persistent FP4 pre-pass fed by an LDS-DMA ring (fp32 tiles with ld % 4 == 0) -----------------------------------
// EXPERIMENT (only in a -DPCOA_EXPERIMENTS build; tools/exp_overlap.hip): bit-identical to and as fast as
// pack_fp4_kernel (profiles/r02a), built to share a CU with the contraction -- which turned out negative-sum: a CU's
// vector-memory path returns in order, so HBM-latency loads beside L2-hit operand loads slow both (profiles/r02d, r02e).
// Same output as pack_fp4_kernel<float, 4>, built to run BESIDE the contraction: 256 threads (one wave per SIMD),
// <= 64 VGPRs and 64 KiB of LDS, i.e. exactly what gram_packed_kernel (2 waves per SIMD x 224 VGPRs, 96 KiB) leaves free
// on a CU, and a fixed grid of ~one workgroup per CU that lives for the whole launch (a stream of short-lived small
// workgroups takes the wave slots a finishing contraction workgroup frees before its successor fits: r01ov1).
// A wave owns units of (k-block, 256 samples) = 32 rows x 1 KiB.  The rows travel HBM -> LDS by global_load_lds_dwordx4
// (1 KiB per instruction, no VGPR staging) into the wave's private ring of R one-row slots; per row the wave reads its
// 16 B back (ds_read_b128, conflict-free), converts 4 values and re-issues the slot R rows ahead, so R - 1 KiB stay in
// flight per wave whatever the registers hold.  Stores share the in-order vmcnt queue with the ring: every wait is
// vmcnt(R - 1), which is correct wherever the four stores of a finished k-block sit in the queue.
// Per value: the nibble is bit 29 of the fp32 pattern (set for 1.0f, clear for 0.0f) shifted into place, and
// fma(f, f, -f) is +0 exactly for f in {0, -0, 1} and non-zero (or NaN) for everything else: 4 VALU ops.
// The 32 rows of one unit.  `a` holds row 0 on entry and row 0 of the wave's next unit on exit.  bad4[s] collects the
// 0/1 check of sample column s; dead columns are masked once per unit by the caller, rows beyond the tile (they re-read
// its last row) are skipped by a wave-uniform branch.
template <int R, int AUX>
__device__ __forceinline__ void ring_unit(const char* xb, int64_t ldb, int nv, int kb, uint32_t voff, int kbn, uint32_t voffn,
                                          uint8_t* myring, int lane, f32x4_t& a, uint32_t (&w)[4][4], uint32_t (&bad4)[4]) {
  auto issue = [&](int kbq, uint32_t vo, int t, int slot) {
    int row = kbq * 32 + t;
    row = row < nv ? row : nv - 1;
    const char* src = xb + (int64_t)row * ldb + vo;
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(myring + slot * 1024), 16, 0, AUX);
  };
#pragma unroll
  for (int t = 0; t < 32; ++t) {
    // (1) row t is in `a` once the ds_read has returned; its slot is free then.  The builtin, not inline asm: the
    // compiler's own wait-count pass sees it and does not add an lgkmcnt(0) of its own in front of the conversion
    // (which would also wait for the NEXT row's read issued in (4))
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0), vmcnt / expcnt untouched
    __builtin_amdgcn_sched_barrier(0);
    // (2) refill the slot with the row R ahead (of this unit, or of the wave's next unit)
    if (t + R < 32) issue(kb, voff, t + R, t % R);
    else issue(kbn, voffn, t + R - 32, t % R);
    // (3) row t+1 has landed when at most R - 1 operations are outstanding
    wait_vmcnt<R - 1>();
    __builtin_amdgcn_sched_barrier(0);
    // (4) its 16 B start on their way from LDS while row t is converted
    const f32x4_t b = *reinterpret_cast<const f32x4_t*>(myring + ((t + 1) % R) * 1024 + lane * 16);
    __builtin_amdgcn_sched_barrier(0);
    // (5) row t
    if (kb * 32 + t < nv) {
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        const float f = a[s2];
        bad4[s2] |= __float_as_uint(__builtin_fmaf(f, f, -f));
        const uint32_t bits = __float_as_uint(f) >> (28 - 4 * (t & 7));
        w[s2][t >> 3] |= bits & (2u << (4 * (t & 7)));
      }
      // pin the conversion here: without a consumer the optimiser sinks all 32 rows' arithmetic behind the loop and
      // keeps every row live in registers
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) asm volatile("" : "+v"(w[s2][t >> 3]), "+v"(bad4[s2]));
    }
    a = b;
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int R, int AUX>
__global__ __launch_bounds__(256, 8) void pack_fp4_ring_kernel(const float* __restrict__ x, int64_t ld, int nv, int n,
                                                               int npad, int n_units, int8_t* __restrict__ p,
                                                               int32_t* __restrict__ flag) {
  static_assert(R == 8 || R == 16 || R == 32, "the slot of row t must be a compile-time constant of the 32-row unrolled body");
  // dynamic LDS (4 * R KiB, passed at launch): with a static 64 KiB array the compiler sees an LDS-limited occupancy
  // of 2 and lets the register allocator spread to 145 VGPRs; the launch bound (8 waves per SIMD = 64 VGPRs) only binds
  // when the LDS size is unknown to it
  extern __shared__ __attribute__((aligned(16))) uint8_t ring_dyn[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gw = npad >> 8;  // 256-sample groups per k-block
  const int stride = (int)gridDim.x * 4;
  const int stride_kb = stride / gw, stride_g = stride - stride_kb * gw;
  int u = (int)blockIdx.x * 4 + wave;
  if (u >= n_units) return;  // no workgroup barrier anywhere below
  int kb = u / gw, G = u - kb * gw;
  uint8_t* const myring = ring_dyn + wave * (R * 1024);
  const char* const xb = reinterpret_cast<const char*>(x);
  const int64_t ldb = ld * 4;
  uint32_t bad = 0;

  // per-lane byte offset inside a row (columns beyond ld read column 0 and are masked later); the row address is
  // wave-uniform
  auto lane_off = [&](int Gq) -> uint32_t {
    const int col = Gq * 256 + lane * 4;
    return col < ld ? (uint32_t)col * 4u : 0u;
  };
  uint32_t voff = lane_off(G);
#pragma unroll
  for (int t = 0; t < R; ++t) {
    int row = kb * 32 + t;
    row = row < nv ? row : nv - 1;
    __builtin_amdgcn_global_load_lds((gptr_t)(xb + (int64_t)row * ldb + voff), (lptr_t)(myring + t * 1024), 16, 0, AUX);
  }
  wait_vmcnt<R - 1>();
  f32x4_t a = *reinterpret_cast<const f32x4_t*>(myring + lane * 16);  // row 0 of the first unit
  for (;;) {
    // the wave's next unit; past the end a harmless re-read of this one keeps the queue depth (and vmcnt) uniform
    int kbn = kb + stride_kb, Gn = G + stride_g;
    if (Gn >= gw) { Gn -= gw; kbn += 1; }
    const bool last = u + stride >= n_units;
    if (last) { kbn = kb; Gn = G; }
    const uint32_t voffn = lane_off(Gn);
    const int col = G * 256 + lane * 4;
    uint32_t w[4][4], bad4[4];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      bad4[s2] = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) w[s2][q] = 0;
    }
    ring_unit<R, AUX>(xb, ldb, nv, kb, voff, kbn, voffn, myring, lane, a, w, bad4);
    // samples >= n (padding columns of the tile, or of the operand) may hold anything: dropped here, once per unit
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      const uint32_t cm = (col + s2 < n) ? 0xffffffffu : 0u;
      bad |= bad4[s2] & cm;
#pragma unroll
      for (int q = 0; q < 4; ++q) w[s2][q] &= cm;
    }
    uint4* dst = reinterpret_cast<uint4*>(p + ((size_t)kb * npad + col) * 16);
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) dst[s2] = make_uint4(w[s2][0], w[s2][1], w[s2][2], w[s2][3]);
    if (last) break;
    u += stride;
    kb = kbn;
    G = Gn;
    voff = voffn;
  }
  wait_vmcnt<0>();  // the ring's tail must have landed before the LDS can belong to another workgroup
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (bad) atomicOr(flag, 8);
}

#endif  // PCOA_EXPERIMENTS

// uint8 input, 8-byte loads: one thread packs 32 variants x 8 samples (a wave reads 512 contiguous bytes per row
// instead of the 256 of the generic kernel above: 2504-byte rows are not line-aligned, so short segments pay for an
// extra 128-B line each).  Four batches of 8 rows; per batch the 0/1 bytes of row t are OR-ed in at bit t, which
// leaves one byte of 8 row-bits per sample, and `spread8` turns that byte into 8 FP4 nibbles (bit t -> 0x2 << 4t).
__device__ __forceinline__ uint32_t spread8_fp4(uint32_t b) {  // b < 256
  uint32_t x = (b | (b << 12)) & 0x000F000Fu;
  x = (x | (x << 6)) & 0x03030303u;
  x = (x | (x << 3)) & 0x11111111u;
  return x << 1;
}

__global__ __launch_bounds__(256) void pack_u8x8_fp4_kernel(const uint8_t* __restrict__ x, int64_t ld, int64_t nv, int n,
                                                            int npad, int64_t nkb_pad, int8_t* __restrict__ p,
                                                            int32_t* __restrict__ flag) {
  const int groups = npad >> 3;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t kb = gid / groups;
  const int g = (int)(gid - kb * groups);
  if (kb >= nkb_pad) return;
  const int i0 = g * 8;
  // byte masks of the columns that exist (< n); columns in [n, ld) may hold anything
  uint32_t m[2];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    uint32_t mm = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if (i0 + 4 * d + b < n) mm |= 0xffu << (8 * b);
    m[d] = mm;
  }
  const bool in_row = i0 < ld;  // ld is a multiple of 8 on this path, so the whole 8-byte load is inside the row
  uint32_t o[8][4];
  uint32_t bad = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    uint32_t a0 = 0, a1 = 0;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int64_t row = kb * 32 + q * 8 + t;
      uint2 u = make_uint2(0u, 0u);
      if (row < nv && in_row) u = *reinterpret_cast<const uint2*>(x + row * ld + i0);
      u.x &= m[0];
      u.y &= m[1];
      bad |= (u.x | u.y) & 0xfefefefeu;
      a0 |= (u.x & 0x01010101u) << t;
      a1 |= (u.y & 0x01010101u) << t;
    }
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
      o[sidx][q] = spread8_fp4((a0 >> (8 * sidx)) & 0xffu);
      o[4 + sidx][q] = spread8_fp4((a1 >> (8 * sidx)) & 0xffu);
    }
  }
  uint4* dst = reinterpret_cast<uint4*>(p + ((size_t)kb * npad + i0) * 16);
#pragma unroll
  for (int sidx = 0; sidx < 8; ++sidx) dst[sidx] = make_uint4(o[sidx][0], o[sidx][1], o[sidx][2], o[sidx][3]);
  if (bad) atomicOr(flag, 8);
}

// ---- bit-packed boundary: carrier bitsets (1 bit per genotype, row v = variant v, bit i & 31 of word i >> 5 =
// sample i) -> P4.  One wave = one k-block (32 variants) x 256 samples: lane (t, j) loads the 16 bytes of row t
// that hold samples 128 j .. 128 j + 127 of the wave's range (one vector load per wave, 32 contiguous bytes per
// row).  The 32 x 32 bit transposes go through v_readlane: the two dwords of row t that belong to a group of 64
// samples land in an SGPR pair, and an SGPR pair IS a lane mask -- v_cndmask_b32 with it as condition hands every
// lane its own sample's bit as a positioned FP4 nibble.  4 VALU ops per (row, 64 samples), one coalesced 16-B store
// per lane and group.  (A first version with scalar loads instead of the vector load + readlane was 5x slower:
// 1.74 ms per 10^6 variants, scalar-cache bound.)  A bitset cannot repeat a callset, so the tile is binary by
// construction and always takes the FP4 kernel.
template <int VEC>
__global__ __launch_bounds__(256) void expand_bits_fp4_kernel(const uint32_t* __restrict__ bits, int64_t ld_words,
                                                              int64_t nv, int n, int npad, int64_t nkb_pad,
                                                              int8_t* __restrict__ p) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int gpk = npad >> 8;                                  // 256-sample groups per k-block
  const int64_t wid = (int64_t)blockIdx.x * 4 + wave;
  const int64_t kb = wid / gpk;
  const int G = (int)(wid - kb * gpk);
  if (kb >= nkb_pad) return;
  const int t = lane & 31, j = lane >> 5;
  const int64_t row = kb * 32 + t;
  const int64_t w = 8 * (int64_t)G + 4 * j;                   // first of this lane's four dwords
  uint32_t c[4] = {0u, 0u, 0u, 0u};
  if (row < nv) {
    const uint32_t* r = bits + row * ld_words;
    if (VEC == 4 && w + 3 < ld_words) {
      const uint4 u = *reinterpret_cast<const uint4*>(r + w);
      c[0] = u.x; c[1] = u.y; c[2] = u.z; c[3] = u.w;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (w + q < ld_words) c[q] = r[w + q];
    }
  }
  // bits of samples >= N (row padding, the tail of the last word) are ignored
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t first = 32 * (w + q);
    if (first >= n) c[q] = 0u;
    else if (first + 32 > n) c[q] &= (1u << (n - (int)first)) - 1u;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {                               // samples 256 G + 64 q + lane
    uint32_t o[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int t2 = 0; t2 < 32; ++t2) {
      const int src = t2 + 32 * (q >> 1);                     // the lane that loaded row t2, half q >> 1
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)c[2 * (q & 1)], src);
      const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)c[2 * (q & 1) + 1], src);
      const uint64_t m = ((uint64_t)hi << 32) | lo;
      uint32_t nib;
      const uint32_t one = 2u << (4 * (t2 & 7));              // E2M1 1.0 at this variant's nibble
      asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(nib) : "v"(one), "s"(m));
      o[t2 >> 3] |= nib;
    }
    *reinterpret_cast<uint4*>(p + ((size_t)kb * npad + (size_t)256 * G + 64 * q + lane) * 16) =
        make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// ---------------------------------------------------------------------------------------------- gemm
// Template parameters
//   NWM  waves along M (tile height 128*NWM); NNI 32-column MFMA tiles per wave along N (wave tile
//        128 x 32*NNI, 8/NNI waves along N); SKB k-blocks (of 16 variants) per stage; NST ring length.
// Measured at N = 2504, 10^6 variants per launch (MI355X):
//   <2,2,4,3> 256x256, 8 waves, 64-variant stages, 3-ring          2.41 ms
//   <1,2,4,3> 128x256, 4 waves, two workgroups per CU              2.88 ms  (more LDS-DMA bytes per MAC)
//   <2,4,4,3> 256x256, 4 waves (one per SIMD), wave tile 128x128   2.61 ms  (LDS latency exposed)
constexpr int TJ = 256;  // tile width (panel J) in samples

template <int NWM, int SKB>
struct StageI8 {
  int8_t pi[SKB][128 * NWM][KB];  // panel I: [k-block][sample][16 B]
  int8_t pj[SKB][TJ][KB];         // panel J
};

// DMA instructions per stage: SKB * (2*NWM + 4) of 1 KiB each, spread evenly over the waves.
template <int NWM, int SKB, int NWAVES>
__device__ __forceinline__ void issue_stage_i8(StageI8<NWM, SKB>* st, const int8_t* __restrict__ p, int npad,
                                               int64_t kb0, int col_i, int col_j, int wave, int lane) {
  constexpr int QI = 2 * NWM;              // 64-sample quarters in panel I
  constexpr int PER_KB = QI + 4;           // instructions per k-block
  constexpr int TOTAL = SKB * PER_KB;
  constexpr int PER_WAVE = TOTAL / NWAVES;
  static_assert(TOTAL % NWAVES == 0, "DMA instructions must divide evenly over the waves");
#pragma unroll
  for (int q = 0; q < PER_WAVE; ++q) {
    const int id = wave * PER_WAVE + q;
    const int kb = id / PER_KB;
    const int r = id - kb * PER_KB;
    const bool is_i = r < QI;
    const int sq = is_i ? r : r - QI;
    const int c0 = (is_i ? col_i : col_j) + sq * 64;
    const int8_t* src = p + ((size_t)(kb0 + kb) * npad + c0 + lane) * KB;
    int8_t* dst = is_i ? &st->pi[kb][sq * 64][0] : &st->pj[kb][sq * 64][0];
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)dst, 16, 0, 0);
  }
}

// Fragment registers of one k32-step: (4 A + NNI B) x 16 B = 24 VGPRs at NNI = 2.
template <int NNI>
struct FragsI8 {
  i32x4 a[4];
  i32x4 b[NNI];
};

template <int NWM, int NNI, int SKB>
__device__ __forceinline__ void load_frags_i8(const StageI8<NWM, SKB>* st, int k2, int wm, int wn, int lane,
                                              FragsI8<NNI>& f) {
  const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
    f.a[mi] = *reinterpret_cast<const i32x4*>(&st->pi[2 * k2 + hi][wm * 128 + mi * 32 + l31][0]);
#pragma unroll
  for (int ni = 0; ni < NNI; ++ni)
    f.b[ni] = *reinterpret_cast<const i32x4*>(&st->pj[2 * k2 + hi][wn * 32 * NNI + ni * 32 + l31][0]);
}

// Diagonal tiles (row block == column block): panel J IS panel I, so only panel I is brought in (16 instead of 32
// DMA instructions per stage) and the B fragments are read from it.
template <int NWM, int NNI, int SKB>
__device__ __forceinline__ void load_frags_diag(const StageI8<NWM, SKB>* st, int k2, int wm, int wn, int lane,
                                                FragsI8<NNI>& f) {
  const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
    f.a[mi] = *reinterpret_cast<const i32x4*>(&st->pi[2 * k2 + hi][wm * 128 + mi * 32 + l31][0]);
#pragma unroll
  for (int ni = 0; ni < NNI; ++ni)
    f.b[ni] = *reinterpret_cast<const i32x4*>(&st->pi[2 * k2 + hi][wn * 32 * NNI + ni * 32 + l31][0]);
}

template <int NWM, int SKB, int NWAVES>
__device__ __forceinline__ void issue_stage_diag(StageI8<NWM, SKB>* st, const int8_t* __restrict__ p, int npad,
                                                 int64_t kb0, int col_i, int wave, int lane) {
  constexpr int QI = 2 * NWM;  // 64-sample quarters in panel I
  constexpr int TOTAL = SKB * QI;
  constexpr int PER_WAVE = TOTAL / NWAVES;
  static_assert(TOTAL % NWAVES == 0, "DMA instructions must divide evenly over the waves");
#pragma unroll
  for (int q = 0; q < PER_WAVE; ++q) {
    const int id = wave * PER_WAVE + q;
    const int kb = id / QI;
    const int sq = id - kb * QI;
    const int8_t* src = p + ((size_t)(kb0 + kb) * npad + col_i + sq * 64 + lane) * KB;
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)&st->pi[kb][sq * 64][0], 16, 0, 0);
  }
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// FMT 0: int8 operands, int32 accumulators.  FMT 1: MX-FP4 operands (scales 2^0), fp32 accumulators.
template <int FMT>
struct AccType { typedef i32x16 type; };
template <>
struct AccType<1> { typedef f32x16 type; };


template <int FMT, int NNI>
__device__ __forceinline__ void mfma_step_i8(const FragsI8<NNI>& f, typename AccType<FMT>::type (&acc)[4][NNI]) {
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < NNI; ++ni) {
      if constexpr (FMT == 0) {
        acc[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f.a[mi], f.b[ni], acc[mi][ni], 0, 0, 0);
      } else {
        // 32 FP4 values per lane = 4 VGPRs; cbsz = blgp = 4 selects E2M1.  The UNSCALED form of the instruction:
        // no block scale is applied (bit-exact against the oracle, tests/test_gpu_parity.py) and it saves the
        // ld_scale half of v_mfma_scale_* (1.235 vs 1.259 ms per 10^6 variants).  Written as inline asm because the builtin takes 8-VGPR operand tuples (the FP4 form
        // reads the low 4): materialising them doubles the fragment registers and the kernel spills.
        // Hazards (cdna_hip_programming.md 5.7): operands come from ds_read (the compiler waits lgkmcnt before
        // the statement since it names them as inputs); D feeds only the next MFMA as its whole C (no wait
        // states needed); the epilogue's first VALU read of D is fenced by s_nops after the main loop.
        asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0 cbsz:4 blgp:4"
                     : "+v"(acc[mi][ni])
                     : "v"(f.a[mi]), "v"(f.b[ni]));
      }
    }
}

// MFMAs number LO .. HI-1 of a stage, in the order (k-step, mi, ni) of mfma_step_i8
template <int FMT, int NNI, int SKB, int LO, int HI>
__device__ __forceinline__ void mfma_range(const FragsI8<NNI> (&f)[SKB / 2], typename AccType<FMT>::type (&acc)[4][NNI]) {
#pragma unroll
  for (int k2 = 0; k2 < SKB / 2; ++k2)
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < NNI; ++ni) {
        const int t = (k2 * 4 + mi) * NNI + ni;
        if (t < LO || t >= HI) continue;
        if constexpr (FMT == 0) {
          acc[mi][ni] = __builtin_amdgcn_mfma_i32_32x32x32_i8(f[k2].a[mi], f[k2].b[ni], acc[mi][ni], 0, 0, 0);
        } else {
          asm volatile("v_mfma_f32_32x32x64_f8f6f4 %0, %1, %2, %0 cbsz:4 blgp:4"
                       : "+v"(acc[mi][ni])
                       : "v"(f[k2].a[mi]), "v"(f[k2].b[ni]));
        }
      }
}

// One stage of the ring.  Prefetch distance D = NST - 1: when stage s is consumed, stages s+1 .. s+D-1
// may still be in flight (counted vmcnt), and stage s+D is issued into the buffer stage s-1 used.
// Inside the stage the fragment reads are software-pipelined one k32-step ahead of the MFMAs
// (two register sets of 24 VGPRs), so the stage depth SKB does not cost registers.
template <int FMT, int NWM, int NNI, int SKB, int NST, int BUF, bool IDLE>
__device__ __forceinline__ void ring_step(StageI8<NWM, SKB>* lds, const int8_t* __restrict__ p, int npad,
                                          int64_t kb_begin, int s, int ns, int col_i, int col_j, int wave, int lane,
                                          int wm, int wn, typename AccType<FMT>::type (&acc)[4][NNI]) {
  constexpr int NWAVES = NWM * (8 / NNI);
  constexpr int PER_WAVE = SKB * (2 * NWM + 4) / NWAVES;
  constexpr int D = NST - 1;
  static_assert(D >= 1 && D <= 5 && D * PER_WAVE < 64, "prefetch distance 1..5, vmcnt is a 6-bit counter");
  {
    // stages s+1 .. s+min(D-1, ns-1-s) may stay in flight
    const int rem = ns - 1 - s;
    const int keep = rem < D - 1 ? rem : D - 1;
    if (keep >= 4) wait_vmcnt<(D >= 5 ? 4 * PER_WAVE : 0)>();
    else if (keep == 3) wait_vmcnt<(D >= 4 ? 3 * PER_WAVE : 0)>();
    else if (keep == 2) wait_vmcnt<(D >= 3 ? 2 * PER_WAVE : 0)>();
    else if (keep == 1) wait_vmcnt<(D >= 2 ? PER_WAVE : 0)>();
    else wait_vmcnt<0>();
  }
  wg_barrier();  // all waves' stage-s DMA landed, and all waves are done reading the buffer of stage s-1
  if constexpr (IDLE) {  // this wave's sub-tile lies below the diagonal: it only helps with the DMA
    if (s + D < ns)
      issue_stage_i8<NWM, SKB, NWAVES>(&lds[(BUF + D) % NST], p, npad, kb_begin + (int64_t)(s + D) * SKB, col_i,
                                       col_j, wave, lane);
    return;
  }
  FragsI8<NNI> f0, f1;
  load_frags_i8<NWM, NNI, SKB>(&lds[BUF], 0, wm, wn, lane, f0);
  __builtin_amdgcn_sched_barrier(0);
  // the DMA of stage s+D is issued under the LDS latency of the first fragment reads
  if (s + D < ns)
    issue_stage_i8<NWM, SKB, NWAVES>(&lds[(BUF + D) % NST], p, npad, kb_begin + (int64_t)(s + D) * SKB, col_i, col_j,
                                     wave, lane);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k2 = 0; k2 < SKB / 2; k2 += 2) {
    if (k2 + 1 < SKB / 2) load_frags_i8<NWM, NNI, SKB>(&lds[BUF], k2 + 1, wm, wn, lane, f1);
    mfma_step_i8<FMT, NNI>(f0, acc);
    if (k2 + 2 < SKB / 2) load_frags_i8<NWM, NNI, SKB>(&lds[BUF], k2 + 2, wm, wn, lane, f0);
    if (k2 + 1 < SKB / 2) mfma_step_i8<FMT, NNI>(f1, acc);
  }
}

template <int NWM, int SKB, int NST, int NWAVES, int... Is>
__device__ __forceinline__ void ring_prologue(StageI8<NWM, SKB>* lds, const int8_t* __restrict__ p, int npad,
                                              int64_t kb_begin, int ns, int col_i, int col_j, int wave, int lane,
                                              std::integer_sequence<int, Is...>) {
  ((Is < ns ? issue_stage_i8<NWM, SKB, NWAVES>(&lds[Is], p, npad, kb_begin + (int64_t)Is * SKB, col_i, col_j, wave,
                                               lane)
            : (void)0),
   ...);
}

// `count` consecutive stages starting at s (s is a multiple of NST, so stage s+i lives in buffer i)
template <int FMT, int NWM, int NNI, int SKB, int NST, bool IDLE, int... Is>
__device__ __forceinline__ void ring_round(StageI8<NWM, SKB>* lds, const int8_t* __restrict__ p, int npad,
                                           int64_t kb_begin, int s, int ns, int count, int col_i, int col_j,
                                           int wave, int lane, int wm, int wn,
                                           typename AccType<FMT>::type (&acc)[4][NNI],
                                           std::integer_sequence<int, Is...>) {
  ((Is < count ? ring_step<FMT, NWM, NNI, SKB, NST, Is, IDLE>(lds, p, npad, kb_begin, s + Is, ns, col_i, col_j, wave, lane,
                                                          wm, wn, acc)
               : (void)0),
   ...);
}

// ---- ping-pong schedule (PP = true) -------------------------------------------------------------------------
// The plain ring above keeps all eight waves in phase: after every barrier they all read fragments, then all
// issue MFMAs, and the matrix pipe idles during the read burst (53-64 % MFMA-busy measured).  Here the waves
// form two groups, one wave of each group per SIMD (waves w and w+4 share a SIMD), half a stage apart:
//
//   phase 2s   : group 0 reads the fragments of stage s (12 ds_read_b128)  | group 1 issues the MFMAs of stage s-1
//   phase 2s+1 : group 0 issues the 16 MFMAs of stage s                    | group 1 reads the fragments of stage s
//
// with one raw s_barrier after every phase, so the matrix pipe of every SIMD always has one wave feeding it while
// the other wave's LDS reads are in flight.  Buffer of stage s is read in phases 2s (group 0) and 2s+1 (group 1) and
// is free after the barrier that ends phase 2s+1; the DMA of stage s+NST-1 into the buffer of stage s-1 is issued by
// both groups at the start of phase 2s and has until the barrier that ends phase 2(s+NST-1)-1 to land (counted
// vmcnt: only the following stage's DMA may still be in flight there).  (Moving group 1's DMA issue out of its
// MFMA phase into its read phase was measured 2 % slower: 1.265 vs 1.237 ms per 10^6 variants, commit 4f657f1.)
__device__ __forceinline__ void raw_barrier() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}

template <int FMT, int NWM, int NNI, int SKB, int NST, int BUF, int GRP, bool IDLE, int LEFT, bool DIAG>
__device__ __forceinline__ void pp_stage(StageI8<NWM, SKB>* lds, const int8_t* __restrict__ p, int npad,
                                         int64_t kb_begin, int s, int ns, int col_i, int col_j, int wave, int lane,
                                         int wm, int wn, typename AccType<FMT>::type (&acc)[4][NNI],
                                         FragsI8<NNI> (&f)[SKB / 2]) {
  constexpr int NWAVES = NWM * (8 / NNI);
  constexpr int PER_WAVE = SKB * (DIAG ? 2 * NWM : 2 * NWM + 4) / NWAVES;
  constexpr int D = NST - 1;
  static_assert(D >= 1 && D * PER_WAVE < 64, "vmcnt is a 6-bit counter");
  // LEFT: the last LEFT MFMAs of a group's run are issued AFTER the barrier that ends its phase (at raised
  // priority), so the matrix pipe has work while the barrier releases and the other group's first MFMA is on its way
  constexpr int TOT = (SKB / 2) * 4 * NNI;
  const bool more = s + D < ns;  // a DMA is issued during this stage
  if constexpr (GRP == 0) {
    // ---- phase 2s: read stage s, issue the DMA of stage s+D
    if constexpr (!IDLE) {
#pragma unroll
      for (int k2 = 0; k2 < SKB / 2; ++k2) {
        if constexpr (DIAG) load_frags_diag<NWM, NNI, SKB>(&lds[BUF], k2, wm, wn, lane, f[k2]);
        else load_frags_i8<NWM, NNI, SKB>(&lds[BUF], k2, wm, wn, lane, f[k2]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
      if constexpr (DIAG)
        issue_stage_diag<NWM, SKB, NWAVES>(&lds[(BUF + D) % NST], p, npad, kb_begin + (int64_t)(s + D) * SKB, col_i, wave,
                                           lane);
      else
        issue_stage_i8<NWM, SKB, NWAVES>(&lds[(BUF + D) % NST], p, npad, kb_begin + (int64_t)(s + D) * SKB, col_i, col_j,
                                         wave, lane);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    raw_barrier();
    // ---- phase 2s+1: the MFMAs of stage s
    if constexpr (!IDLE) {
      __builtin_amdgcn_s_setprio(1);
      mfma_range<FMT, NNI, SKB, 0, TOT - LEFT>(f, acc);
      __builtin_amdgcn_s_setprio(0);
    }
  } else {
    // ---- phase 2s: issue the DMA of stage s+D, then the MFMAs of stage s-1
    if (more) {
      if constexpr (DIAG)
        issue_stage_diag<NWM, SKB, NWAVES>(&lds[(BUF + D) % NST], p, npad, kb_begin + (int64_t)(s + D) * SKB, col_i, wave,
                                           lane);
      else
        issue_stage_i8<NWM, SKB, NWAVES>(&lds[(BUF + D) % NST], p, npad, kb_begin + (int64_t)(s + D) * SKB, col_i, col_j,
                                         wave, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!IDLE) {
      if (s > 0) {
        __builtin_amdgcn_s_setprio(1);
        mfma_range<FMT, NNI, SKB, 0, TOT - LEFT>(f, acc);
        __builtin_amdgcn_s_setprio(0);
      }
    }
    raw_barrier();
    // ---- phase 2s+1: (the leftover MFMAs of stage s-1, then) read stage s
    if constexpr (!IDLE) {
      if constexpr (LEFT > 0) {
        if (s > 0) {
          __builtin_amdgcn_s_setprio(2);
          mfma_range<FMT, NNI, SKB, TOT - LEFT, TOT>(f, acc);
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int k2 = 0; k2 < SKB / 2; ++k2) {
        if constexpr (DIAG) load_frags_diag<NWM, NNI, SKB>(&lds[BUF], k2, wm, wn, lane, f[k2]);
        else load_frags_i8<NWM, NNI, SKB>(&lds[BUF], k2, wm, wn, lane, f[k2]);
      }
    }
  }
  // end of phase 2s+1: stage s+1 must have landed (own share), only the DMA of stage s+2.. may stay in flight
  if (s + 1 < ns) {
    const int rem = ns - 2 - s;  // stages after s+1 whose DMA has been issued: min(rem, D-1)
    const int keep = rem < D - 1 ? rem : D - 1;
    if (keep >= 2) wait_vmcnt<(D >= 3 ? 2 * PER_WAVE : 0)>();
    else if (keep == 1) wait_vmcnt<(D >= 2 ? PER_WAVE : 0)>();
    else wait_vmcnt<0>();
  }
  if constexpr (GRP == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  raw_barrier();
  if constexpr (GRP == 0 && !IDLE && LEFT > 0) {  // group 0's leftover MFMAs of stage s, into phase 2(s+1)
    __builtin_amdgcn_s_setprio(2);
    mfma_range<FMT, NNI, SKB, TOT - LEFT, TOT>(f, acc);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <int FMT, int NWM, int NNI, int SKB, int NST, int GRP, bool IDLE, int LEFT, bool DIAG, int... Is>
__device__ __forceinline__ void pp_round(StageI8<NWM, SKB>* lds, const int8_t* __restrict__ p, int npad,
                                         int64_t kb_begin, int s, int ns, int count, int col_i, int col_j, int wave,
                                         int lane, int wm, int wn, typename AccType<FMT>::type (&acc)[4][NNI],
                                         FragsI8<NNI> (&f)[SKB / 2], std::integer_sequence<int, Is...>) {
  ((Is < count ? pp_stage<FMT, NWM, NNI, SKB, NST, Is, GRP, IDLE, LEFT, DIAG>(lds, p, npad, kb_begin, s + Is, ns, col_i, col_j, wave,
                                                                  lane, wm, wn, acc, f)
               : (void)0),
   ...);
}

template <int FMT, int NWM, int NNI, int SKB, int NST, int GRP, bool IDLE, int LEFT, bool DIAG>
__device__ __forceinline__ void pp_loop(StageI8<NWM, SKB>* lds, const int8_t* __restrict__ p, int npad,
                                        int64_t kb_begin, int ns, int col_i, int col_j, int wave, int lane, int wm,
                                        int wn, typename AccType<FMT>::type (&acc)[4][NNI]) {
  FragsI8<NNI> f[SKB / 2];
  // prologue: stages 0 .. NST-2 go in flight; stage 0 must have landed before group 0 reads it in phase 0
  constexpr int NWAVES = NWM * (8 / NNI);
  constexpr int PER_WAVE = SKB * (DIAG ? 2 * NWM : 2 * NWM + 4) / NWAVES;
#pragma unroll
  for (int i = 0; i < NST - 1; ++i) {
    if (i < ns) {
      if constexpr (DIAG)
        issue_stage_diag<NWM, SKB, NWAVES>(&lds[i], p, npad, kb_begin + (int64_t)i * SKB, col_i, wave, lane);
      else
        issue_stage_i8<NWM, SKB, NWAVES>(&lds[i], p, npad, kb_begin + (int64_t)i * SKB, col_i, col_j, wave, lane);
    }
  }
  if (ns > 1 && NST > 2) wait_vmcnt<(NST > 2 ? PER_WAVE : 0)>();
  else wait_vmcnt<0>();
  raw_barrier();
  int s = 0;
  for (; s + NST - 1 < ns; s += NST)
    pp_round<FMT, NWM, NNI, SKB, NST, GRP, IDLE, LEFT, DIAG>(lds, p, npad, kb_begin, s, ns, NST, col_i, col_j, wave, lane, wm, wn, acc,
                                                 f, std::make_integer_sequence<int, NST>{});
  if (s < ns)
    pp_round<FMT, NWM, NNI, SKB, NST, GRP, IDLE, LEFT, DIAG>(lds, p, npad, kb_begin, s, ns, ns - s, col_i, col_j, wave, lane, wm, wn,
                                                 acc, f, std::make_integer_sequence<int, NST - 1>{});
  if constexpr (GRP == 1 && !IDLE) {  // phase 2*ns: group 1's MFMAs of the last stage, nobody to wait for
#pragma unroll
    for (int k2 = 0; k2 < SKB / 2; ++k2) mfma_step_i8<FMT, NNI>(f[k2], acc);
  }
}

// Tile enumeration over the upper triangle (any bijection is valid: every tile is computed once).
//   NWM = 2: tiles (ti <= tj) of 256 x 256, visited in BANDS of 16 tile rows; inside a band the order is
//            column by column.  Workgroups that run at the same time (~256 consecutive indices) then
//            cover about a 16 x 16 block of tiles and share 16 + 16 operand panels instead of 1 + 256,
//            which is what keeps the contraction off the HBM roofline when N is large (N = 100k: 77,028
//            tiles, 4 MB of operand per panel and launch).  At N = 2504 there is a single band.
//   NWM = 1: row block r in [0, 2T) of 128 samples, column block c >= r/2: T(T+1) tiles, simple order.
constexpr int BAND = 16;

template <int NWM>
__device__ __forceinline__ void tile_coords(int tile, int ntile, int& row_blk, int& col_blk) {
  if (NWM == 2 && ntile <= BAND) {
    // a single band: plain row-major order (measured at T = 10: 36 % fewer HBM fetches than column order)
    int ti = 0, rem = tile;
    while (rem >= ntile - ti) {
      rem -= ntile - ti;
      ++ti;
    }
    row_blk = ti;
    col_blk = ti + rem;
  } else if (NWM == 2) {
    int r0 = 0, rem = tile;
    for (;;) {
      const int h = (ntile - r0 < BAND) ? (ntile - r0) : BAND;   // rows in this band
      const int in_band = h * (h + 1) / 2 + (ntile - r0 - h) * h;
      if (rem < in_band) {
        const int tri = h * (h + 1) / 2;
        if (rem < tri) {               // triangular head: column c (relative) holds c + 1 tiles
          int c = 0;
          while (rem >= c + 1) {
            rem -= c + 1;
            ++c;
          }
          row_blk = r0 + rem;
          col_blk = r0 + c;
        } else {                       // rectangular part: h tiles per column
          const int q = rem - tri;
          row_blk = r0 + q % h;
          col_blk = r0 + h + q / h;
        }
        return;
      }
      rem -= in_band;
      r0 += h;
    }
  } else {
    int sup = 0, rem = tile;
    while (rem >= 2 * (ntile - sup)) {
      rem -= 2 * (ntile - sup);
      ++sup;
    }
    const int half = rem / (ntile - sup);
    row_blk = 2 * sup + half;
    col_blk = sup + (rem - half * (ntile - sup));
  }
}

#ifdef PCOA_EXPERIMENTS
// Experiment (xcd_map = 3; 10 x 10 tile triangle, two k-streams): the 55 tiles dealt to the 4 XCDs of a k-stream so
// that every XCD touches exactly 6 of the 10 operand panels (a covering design on the panel pairs {0,1} .. {8,9}:
// ABC, ADE, BDE, CDE) instead of 10 / 9 / 7 / 5 with the row-major cut.  Entries are 10 * row_blk + col_blk, -1 = idle.
__device__ const signed char kBalancedTiles[4][14] = {
    {2, 3, 12, 13, 4, 5, 14, 15, 24, 25, 34, 35, 0, 11},
    {6, 7, 16, 17, 8, 9, 18, 19, 1, 66, 67, 77, 68, 69},
    {26, 27, 36, 37, 28, 29, 38, 39, 22, 23, 33, 78, 79, 88},
    {46, 47, 56, 57, 48, 49, 58, 59, 44, 45, 55, 89, 99, -1}};
int g_lockstep_map = 2;   // harness knob: 3 selects the balanced deal where it applies
#endif

template <int FMT, int NWM, int NNI, int SKB, int NST, bool PP, int LEFT = 0>
__global__ __launch_bounds__(64 * NWM * (8 / NNI), (NNI == 2) ? 2 : 1) void gram_packed_kernel(
    const int8_t* __restrict__ p, int npad, int64_t nstages, int n, int ntile, int ntri, int splitk,
    int64_t stages_per, int32_t* __restrict__ s32, int xcd_map, const int32_t* __restrict__ skip, GramStrip strip) {
  __shared__ __attribute__((aligned(16))) StageI8<NWM, SKB> lds[NST];
  // device-side predicate of the auto mode: a pre-pass met a value other than 0 / 1 in the buffered tiles, so this
  // launch must not add anything to S (the host redoes those tiles on the int8 kernel once it reads the same word)
  if (skip != nullptr && *skip != 0) return;

  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int NWN = 8 / NNI;  // waves along N
  constexpr int NWAVES = NWM * NWN;
  const int wm = wave / NWN, wn = wave % NWN;

  int tile, ks;
  int row_blk = -1, col_blk = -1;
  const int b = blockIdx.x;
#ifdef PCOA_EXPERIMENTS
  if (xcd_map == 3) {
    const int xcd = b & 7, slot = b >> 3;
    if (slot >= 14) return;
    const int code = kBalancedTiles[xcd & 3][slot];
    if (code < 0) return;
    row_blk = code / 10;
    col_blk = code % 10;
    ks = xcd >> 2;
    tile = 0;
  } else
#endif
  if (xcd_map == 2) {
    // lock-step layout: the chip holds ALL tiles of `splitk` k-streams at once, one workgroup per CU for the whole
    // launch.  A k-stream's tiles live on a group of 8 / splitk XCDs, so every operand row is fetched into those L2s
    // once and shared by workgroups that move through k together (no second round that re-reads the slice).
    const int g = kNumXcd / splitk;               // XCDs per k-stream
    const int per = (ntri + g - 1) / g;           // tiles per XCD
    const int xcd = b & 7, slot = b >> 3;
    tile = (xcd % g) * per + slot;
    ks = xcd / g;
    if (slot >= per || tile >= ntri) return;
  } else if (xcd_map) {
    const int q = b >> 3;
    ks = (b & 7) + kNumXcd * (q / ntri);
    tile = q % ntri;
  } else {
    tile = b % ntri;
    ks = b / ntri;
  }
  if (row_blk >= 0) {
    // coordinates already set (experiment deal)
  } else if (strip.cols > 0) {
    // strip owner: ALL tiles (row block, column block of the strip), in BANDS of 16 tile rows, column by column inside a
    // band -- workgroups that run together then cover ~16 x 16 tiles and share 16 + 16 operand panels (the ordering
    // that keeps the symmetric job MFMA-bound at N = 100,000, tile_coords)
    const int ctiles = ntri / ntile;
    const int per_band = BAND * ctiles;
    const int band = tile / per_band;
    const int r0 = band * BAND;
    const int h = (ntile - r0 < BAND) ? (ntile - r0) : BAND;
    const int rem = tile - band * per_band;
    row_blk = r0 + rem % h;
    col_blk = strip.cb0 + rem / h;
  } else {
    tile_coords<NWM>(tile, ntile, row_blk, col_blk);
  }

  const int64_t st_begin = (int64_t)ks * stages_per;
  const int64_t st_end = (st_begin + stages_per < nstages) ? (st_begin + stages_per) : nstages;
  if (st_begin >= st_end) return;
  const int ns = (int)(st_end - st_begin);
  const int64_t kb_begin = st_begin * SKB;
  const int col_i = row_blk * 128 * NWM, col_j = col_blk * TJ;
  // Diagonal tiles: a wave whose whole 128 x (32*NNI) sub-tile has row > column holds nothing of the
  // upper triangle; it skips its MFMAs (2 of 8 waves in 10 of 55 tiles at N = 2504).
  // (a strip owner keeps both triangles: nothing is idle there)
  const bool idle = strip.cols == 0 && (col_i + wm * 128) > (col_j + wn * 32 * NNI + 32 * NNI - 1);

  typename AccType<FMT>::type acc[4][NNI];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < NNI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0;

  if constexpr (PP) {
    static_assert(!PP || (NWM == 2 && NNI == 2), "ping-pong: 8 waves, group = wave / 4 = wm");
    if (row_blk == col_blk) {  // diagonal tile: one panel, below-diagonal waves idle (workgroup-uniform branch)
      if (wm == 0) {
        pp_loop<FMT, NWM, NNI, SKB, NST, 0, false, LEFT, true>(lds, p, npad, kb_begin, ns, col_i, col_j, wave, lane, wm, wn, acc);
      } else if (idle) {
        pp_loop<FMT, NWM, NNI, SKB, NST, 1, true, LEFT, true>(lds, p, npad, kb_begin, ns, col_i, col_j, wave, lane, wm, wn, acc);
        return;
      } else {
        pp_loop<FMT, NWM, NNI, SKB, NST, 1, false, LEFT, true>(lds, p, npad, kb_begin, ns, col_i, col_j, wave, lane, wm, wn, acc);
      }
    } else if (wm == 0) {
      pp_loop<FMT, NWM, NNI, SKB, NST, 0, false, LEFT, false>(lds, p, npad, kb_begin, ns, col_i, col_j, wave, lane, wm, wn, acc);
    } else {
      pp_loop<FMT, NWM, NNI, SKB, NST, 1, false, LEFT, false>(lds, p, npad, kb_begin, ns, col_i, col_j, wave, lane, wm, wn, acc);
    }
  } else {
  // prologue: stages 0 .. D-1 go in flight
  ring_prologue<NWM, SKB, NST, NWAVES>(lds, p, npad, kb_begin, ns, col_i, col_j, wave, lane,
                                      std::make_integer_sequence<int, NST - 1>{});
  int s = 0;
  if (idle) {  // wave-uniform: a separate loop with no accumulator traffic at all, then nothing to store
    for (; s + NST - 1 < ns; s += NST)
      ring_round<FMT, NWM, NNI, SKB, NST, true>(lds, p, npad, kb_begin, s, ns, NST, col_i, col_j, wave, lane, wm, wn, acc,
                                           std::make_integer_sequence<int, NST>{});
    if (s < ns)
      ring_round<FMT, NWM, NNI, SKB, NST, true>(lds, p, npad, kb_begin, s, ns, ns - s, col_i, col_j, wave, lane, wm, wn,
                                           acc, std::make_integer_sequence<int, NST - 1>{});
    return;
  }
  for (; s + NST - 1 < ns; s += NST)
    ring_round<FMT, NWM, NNI, SKB, NST, false>(lds, p, npad, kb_begin, s, ns, NST, col_i, col_j, wave, lane, wm, wn, acc,
                                          std::make_integer_sequence<int, NST>{});
  if (s < ns)
    ring_round<FMT, NWM, NNI, SKB, NST, false>(lds, p, npad, kb_begin, s, ns, ns - s, col_i, col_j, wave, lane, wm, wn,
                                          acc, std::make_integer_sequence<int, NST - 1>{});
  }

  if constexpr (FMT >= 1) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // last asm MFMA -> VALU read of D
  // epilogue: C/D layout col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
  // Only the upper triangle (j >= i) is authoritative; pcoa_gram_finalize mirrors it.
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
#pragma unroll
    for (int ni = 0; ni < NNI; ++ni) {
      const int j = col_j + wn * 32 * NNI + ni * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = col_i + wm * 128 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int v = (int)acc[mi][ni][r];  // fp32 accumulators hold exact integers below 2^24
        if (strip.cols > 0) {           // S[:, col0 .. col0 + cols): every row, the strip's columns, row length `cols`
          if (i < n && j >= strip.col0 && j < strip.col0 + strip.cols && v != 0)
            atomicAdd(&s32[(int64_t)i * strip.cols + (j - strip.col0)], v);
        } else if (j >= i && j < n && v != 0) {
          atomicAdd(&s32[(int64_t)i * n + j], v);
        }
      }
      // keep the float->int conversions of one MFMA tile next to their atomics: hoisting all 128 of them
      // ahead of the stores would need 128 more registers (the FP4 variant then spills)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
}

#define PCOA_KBITS_KERNELS
#include "gram_kbits.inl"
#undef PCOA_KBITS_KERNELS
#define PCOA_KBITS_W4_KERNELS
#include "gram_kbits_w4.inl"
#undef PCOA_KBITS_W4_KERNELS

}  // namespace

int64_t gram_packed_npad(int32_t n) { return ((int64_t)n + TM - 1) / TM * TM; }
// k-blocks (16 variants for int8, 32 for FP4; 16 B per sample either way) are padded to a multiple of 24 so
// that every stage depth (4, 6 or 8 k-blocks) divides the count
int64_t gram_kb_pad(int64_t nv, int fmt) {
  const int per = fmt >= 1 ? 32 : KB;
  const int64_t nkb = (nv + per - 1) / per;
  return (nkb + 23) / 24 * 24;
}
int64_t gram_packed_kb_pad_i8(int64_t nv) { return gram_kb_pad(nv, 0); }
size_t gram_packed_workspace_bytes(int32_t n, int64_t nv) {  // the int8 size also covers the (half as large) FP4 operand
  return (size_t)gram_packed_kb_pad_i8(nv) * (size_t)gram_packed_npad(n) * KB;
}

// nkb_out: k-blocks to write (the tail beyond nv is zero-filled); 0 = the padded count gram_kb_pad(nv, 1)
hipError_t launch_pack_fp4(const void* x, int is_u8, int64_t ld, int64_t nv, int32_t n, int8_t* p, int32_t* flag,
                           hipStream_t stream, int64_t nkb_out) {
  if (nv <= 0) return hipSuccess;
  const int npad = (int)gram_packed_npad(n);
  const int64_t nkb_pad = nkb_out > 0 ? nkb_out : gram_kb_pad(nv, 1);
  const int64_t threads = nkb_pad * (npad >> 2);
  const int64_t blocks = (threads + 255) / 256;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  const uintptr_t addr = reinterpret_cast<uintptr_t>(x);
  const dim3 grid((unsigned)blocks), block(256);
  if (is_u8) {
    const bool vec = ((ld & 3) == 0) && ((addr & 3) == 0);
    const uint8_t* xs = static_cast<const uint8_t*>(x);
    if (((ld & 7) == 0) && ((addr & 7) == 0)) {
      const int64_t blocks8 = (nkb_pad * (npad >> 3) + 255) / 256;
      if (blocks8 > 0x7fffffffLL) return hipErrorInvalidValue;
      hipLaunchKernelGGL(pack_u8x8_fp4_kernel, dim3((unsigned)blocks8), block, 0, stream, xs, ld, nv, n, npad, nkb_pad, p,
                         flag);
      return hipGetLastError();
    }
    if (vec) hipLaunchKernelGGL((pack_fp4_kernel<uint8_t, 4>), grid, block, 0, stream, xs, ld, nv, n, npad, nkb_pad, p, flag);
    else hipLaunchKernelGGL((pack_fp4_kernel<uint8_t, 1>), grid, block, 0, stream, xs, ld, nv, n, npad, nkb_pad, p, flag);
  } else {
    const bool vec = ((ld & 3) == 0) && ((addr & 15) == 0);
    const float* xs = static_cast<const float*>(x);
    // the fp32 tile is streamed once: nontemporal loads (measured 2.07 vs 2.15 ms per 10^6 variants)
    if (vec) hipLaunchKernelGGL((pack_fp4_kernel<float, 4, true>), grid, block, 0, stream, xs, ld, nv, n, npad, nkb_pad, p, flag);
    else hipLaunchKernelGGL((pack_fp4_kernel<float, 1>), grid, block, 0, stream, xs, ld, nv, n, npad, nkb_pad, p, flag);
  }
  return hipGetLastError();
}

// Persistent LDS-ring twin of the fp32 FP4 pre-pass (pack_fp4_ring_kernel): `wgs` workgroups of 4 waves walk the
// nkb_out x (Npad / 256) units.  Needs ld % 4 == 0 and a 16-byte aligned tile (the caller falls back to launch_pack_fp4).
// (fp32 tiles the vectorised pre-passes take: ld % 4 == 0 and a 16-byte aligned base)
bool pack_fp4_ring_ok(const void* x, int64_t ld) {
  return ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
}
#ifdef PCOA_EXPERIMENTS
hipError_t launch_pack_fp4_ring(const float* x, int64_t ld, int64_t nv, int32_t n, int8_t* p, int32_t* flag,
                                hipStream_t stream, int64_t nkb_out, int wgs, int nt) {
  if (nv <= 0) return hipSuccess;
  const int npad = (int)gram_packed_npad(n);
  const int64_t nkb = nkb_out > 0 ? nkb_out : gram_kb_pad(nv, 1);
  const int64_t n_units = nkb * (npad >> 8);
  if (n_units > 0x3fffffffLL || nv > 0x3fffffffLL) return hipErrorInvalidValue;  // 32-bit unit / row indices in the kernel
  int64_t blocks = (n_units + 3) / 4;
  if (wgs > 0 && blocks > wgs) blocks = wgs;
  // nt: 0 default cache policy, 1 nontemporal; bit 1 of `nt` selects the 8-slot ring (32 KiB of LDS instead of 64)
  const bool small = (nt & 2) != 0;
#define PCOA_RING(R_, AUX_)                                                                                              \
  hipLaunchKernelGGL((pack_fp4_ring_kernel<R_, AUX_>), dim3((unsigned)blocks), dim3(256), 4 * R_ * 1024, stream, x, ld, \
                     (int)nv, n, npad, (int)n_units, p, flag)
  if (small) { if (nt & 1) PCOA_RING(8, 2); else PCOA_RING(8, 0); }
  else { if (nt & 1) PCOA_RING(16, 2); else PCOA_RING(16, 0); }
#undef PCOA_RING
  return hipGetLastError();
}
#endif  // PCOA_EXPERIMENTS

hipError_t launch_expand_bits_fp4(const uint32_t* bits, int64_t ld_words, int64_t nv, int32_t n, int8_t* p,
                                  hipStream_t stream, int64_t nkb_out) {
  if (nv <= 0) return hipSuccess;
  const int npad = (int)gram_packed_npad(n);
  const int64_t nkb_pad = nkb_out > 0 ? nkb_out : gram_kb_pad(nv, 1);
  const int64_t blocks = (nkb_pad * (npad >> 8) + 3) / 4;  // 4 waves per block, one (k-block, 256 samples) each
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  const bool vec = ((ld_words & 3) == 0) && ((reinterpret_cast<uintptr_t>(bits) & 15) == 0);
  if (vec)
    hipLaunchKernelGGL(expand_bits_fp4_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, stream, bits, ld_words, nv, n,
                       npad, nkb_pad, p);
  else
    hipLaunchKernelGGL(expand_bits_fp4_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, bits, ld_words, nv, n,
                       npad, nkb_pad, p);
  return hipGetLastError();
}

hipError_t launch_pack_f32_i8(const float* x, int64_t ld, int64_t nv, int32_t n, int8_t* p, int32_t* flag,
                              hipStream_t stream) {
  if (nv <= 0) return hipSuccess;
  const int npad = (int)gram_packed_npad(n);
  const int64_t nkb_pad = gram_packed_kb_pad_i8(nv);
  const int64_t threads = nkb_pad * (npad >> 2);
  const int64_t blocks = (threads + 255) / 256;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  const bool vec4 = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  if (vec4)
    hipLaunchKernelGGL(pack_f32_i8_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, stream, x, ld, nv, n, npad,
                       nkb_pad, p, flag);
  else
    hipLaunchKernelGGL(pack_f32_i8_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, x, ld, nv, n, npad,
                       nkb_pad, p, flag);
  return hipGetLastError();
}

hipError_t launch_pack_u8_i8(const uint8_t* x, int64_t ld, int64_t nv, int32_t n, int8_t* p, int32_t* flag,
                             hipStream_t stream) {
  if (nv <= 0) return hipSuccess;
  const int npad = (int)gram_packed_npad(n);
  const int64_t nkb_pad = gram_packed_kb_pad_i8(nv);
  const int64_t threads = nkb_pad * (npad >> 2);
  const int64_t blocks = (threads + 255) / 256;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  const int vec_ok = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 3) == 0);
  hipLaunchKernelGGL(pack_u8_i8_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, ld, nv, n, npad, nkb_pad, p,
                     flag, vec_ok);
  return hipGetLastError();
}

hipError_t launch_densify_csr_i8(const int32_t* idx_dev, const int64_t* offs_dev, int64_t nv, int64_t offs_base,
                                 int8_t* p, int32_t n, int32_t* flag, hipStream_t stream) {
  if (nv <= 0) return hipSuccess;
  hipError_t e = hipMemsetAsync(p, 0, gram_packed_workspace_bytes(n, nv), stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(densify_csr_i8_kernel, dim3((unsigned)((nv + 3) / 4)), dim3(256), 0, stream, idx_dev, offs_dev,
                     nv, offs_base, p, (int)gram_packed_npad(n), n, flag);
  return hipGetLastError();
}

// carrier lists without repeated callsets -> nkb_out = ceil(nv / 32) k-blocks of FP4 operand at p
hipError_t launch_densify_csr_fp4(const int32_t* idx_dev, const int64_t* offs_dev, int64_t nv, int64_t offs_base,
                                  int8_t* p, int32_t n, int32_t* flag, hipStream_t stream, int64_t nkb_out) {
  if (nv <= 0) return hipSuccess;
  const int npad = (int)gram_packed_npad(n);
  hipError_t e = hipMemsetAsync(p, 0, (size_t)nkb_out * (size_t)npad * 16, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(densify_csr_fp4_kernel, dim3((unsigned)((nv + 3) / 4)), dim3(256), 0, stream, idx_dev, offs_dev, nv,
                     offs_base, p, npad, n, flag);
  return hipGetLastError();
}

hipError_t launch_gram_i8_packed(const int8_t* p, int64_t nv, int32_t n, int32_t* s32, int num_cu,
                                 hipStream_t stream, int* splitk_out) {
  return launch_gram_packed(p, 0, nv, n, s32, num_cu, stream, splitk_out);
}

// Lock-step launch of the FP4 / int8 contraction (gram_packed_kernel with xcd_map = 2): ntri * splitk <= #CUs
// persistent workgroups, splitk in {1, 2, 4, 8} k-streams, each on 8 / splitk XCDs (DESIGN_HISTORY.md 4.1).
int gram_lockstep_splitk(int32_t n, int cus) {
  const int ntile = (int)(gram_packed_npad(n) / TJ);
  const int64_t ntri = (int64_t)ntile * (ntile + 1) / 2;
  if (cus < kNumXcd) return 0;
  for (int k : {8, 4, 2, 1}) {
    const int g = kNumXcd / k;
    const int64_t per = (ntri + g - 1) / g;
    if (per * kNumXcd <= cus) return k;   // one workgroup per CU, cus / 8 CUs per XCD
  }
  return 0;
}
int gram_lockstep_workgroups(int32_t n, int splitk) {
  const int ntile = (int)(gram_packed_npad(n) / TJ);
  return ntile * (ntile + 1) / 2 * splitk;
}
hipError_t launch_gram_packed_lockstep(const int8_t* p, int fmt, int64_t nv, int32_t n, int32_t* s32, int num_cu,
                                       hipStream_t stream, const int32_t* skip) {
  if (nv <= 0) return hipSuccess;
  const int splitk = gram_lockstep_splitk(n, num_cu > 0 ? num_cu : 256);
  if (splitk == 0) return hipErrorInvalidValue;
  const int npad = (int)gram_packed_npad(n);
  const int ntile = npad / TJ;
  const int ntri = ntile * (ntile + 1) / 2;
  const int skb = 4;
  const int64_t nstages = gram_kb_pad(nv, fmt) / skb;
  const int64_t stages_per = (nstages + splitk - 1) / splitk;
  const int g = kNumXcd / splitk;
  const int per = (ntri + g - 1) / g;
  const dim3 grid((unsigned)(per * kNumXcd)), block(512);
  int map = 2;
#ifdef PCOA_EXPERIMENTS
  if (g_lockstep_map == 3 && ntile == 10 && splitk == 2) map = 3;
#endif
  if (fmt == 1)
    hipLaunchKernelGGL((gram_packed_kernel<1, 2, 2, 4, 3, true, 2>), grid, block, 0, stream, p, npad, nstages, n, ntile, ntri,
                       splitk, stages_per, s32, map, skip, GramStrip{});
  else
    hipLaunchKernelGGL((gram_packed_kernel<0, 2, 2, 4, 3, true, 2>), grid, block, 0, stream, p, npad, nstages, n, ntile, ntri,
                       splitk, stages_per, s32, map, skip, GramStrip{});
  return hipGetLastError();
}

#define PCOA_KBITS_LAUNCHERS
#include "gram_kbits.inl"
#undef PCOA_KBITS_LAUNCHERS
#define PCOA_KBITS_W4_LAUNCHERS
#include "gram_kbits_w4.inl"
#undef PCOA_KBITS_W4_LAUNCHERS

hipError_t launch_gram_packed(const int8_t* p, int fmt, int64_t nv, int32_t n, int32_t* s32, int num_cu,
                              hipStream_t stream, int* splitk_out, const int32_t* skip, GramStrip strip) {
  if (nv <= 0) return hipSuccess;
  // Shipped schedule: ping-pong, 4 k-blocks per stage, 3-stage ring, two MFMAs behind the phase barrier ("243").  The
  // other schedules of DESIGN_HISTORY.md 4.1 / 4.2 (PCOA_GRAM_I8_CFG = 43 | 44 | 143 | 144 | 443) only exist in a library built
  // with -DPCOA_EXPERIMENTS.
  int cfg = 243;
#ifdef PCOA_EXPERIMENTS
  {
    const int t = debug_knobs().gram_cfg;
    if (t == 43 || t == 44 || t == 143 || t == 144 || t == 443) cfg = t;
  }
#endif
  const int skb = (cfg % 100) / 10;
  const int npad = (int)gram_packed_npad(n);
  const int ntile = npad / TJ;
  int64_t ntri64 = (int64_t)ntile * (ntile + 1) / 2;
  if (strip.cols > 0) {  // strip owner: ntile row blocks x the column blocks that touch [col0, col0 + cols)
    strip.cb0 = strip.col0 / TJ;
    const int cb1 = (strip.col0 + strip.cols + TJ - 1) / TJ;
    ntri64 = (int64_t)ntile * (cb1 - strip.cb0);
  }
  if (ntri64 > (1 << 28)) return hipErrorInvalidValue;
  const int ntri = (int)ntri64;
  const int64_t nstages = gram_kb_pad(nv, fmt) / skb;
  // one 512-thread workgroup per CU is resident; aim at ~7 work units per CU, >= 1024 variants each
  const int64_t target = (int64_t)(num_cu > 0 ? num_cu : 256) * 7;
  int64_t splitk = (target + ntri - 1) / ntri;
  if (debug_knobs().gram_splitk > 0) splitk = debug_knobs().gram_splitk;  // experiment hook
  const int64_t max_by_work = nstages * skb / 64;
  if (splitk > max_by_work) splitk = max_by_work;
  if (splitk < 1) splitk = 1;
  int xcd_map = 0;
  if (splitk >= kNumXcd) {
    splitk = (splitk / kNumXcd) * kNumXcd;
    xcd_map = 1;
  }
  const int64_t stages_per = (nstages + splitk - 1) / splitk;
  const int64_t nblocks = (int64_t)ntri * splitk;
  if (nblocks > 0x7fffffffLL) return hipErrorInvalidValue;
  if (splitk_out) *splitk_out = (int)splitk;
  const dim3 grid((unsigned)nblocks), block(512);
#define PCOA_LAUNCH_I8(SKB_, NST_, PP_, LEFT_)                                                                            \
  do {                                                                                                          \
    if (fmt == 1)                                                                                               \
      hipLaunchKernelGGL((gram_packed_kernel<1, 2, 2, SKB_, NST_, PP_, LEFT_>), grid, block, 0, stream, p, npad, nstages, n, ntile, \
                         ntri, (int)splitk, stages_per, s32, xcd_map, skip, strip);                             \
    else                                                                                                        \
      hipLaunchKernelGGL((gram_packed_kernel<0, 2, 2, SKB_, NST_, PP_, LEFT_>), grid, block, 0, stream, p, npad, nstages, n, ntile, \
                         ntri, (int)splitk, stages_per, s32, xcd_map, skip, strip);                             \
  } while (0)
  switch (cfg) {
#ifdef PCOA_EXPERIMENTS
    case 44: PCOA_LAUNCH_I8(4, 4, false, 0); break;
    case 144: PCOA_LAUNCH_I8(4, 4, true, 0); break;
    case 43: PCOA_LAUNCH_I8(4, 3, false, 0); break;
    case 143: PCOA_LAUNCH_I8(4, 3, true, 0); break;
    case 443: PCOA_LAUNCH_I8(4, 3, true, 4); break;
#endif
    default: PCOA_LAUNCH_I8(4, 3, true, 2); break;  // measured: 1.218 (LEFT 2) / 1.218 (4) / 1.236 (0) ms per 10^6 variants
  }
#undef PCOA_LAUNCH_I8
  return hipGetLastError();
}

}  // namespace pcoa
