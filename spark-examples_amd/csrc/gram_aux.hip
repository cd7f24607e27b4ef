// gram_aux.hip -- the small HBM-bound kernels around the Gram contraction:
//   densify_csr      RDD[Seq[Int]] carrier lists (getCallsRdd, VariantsPca.scala:153-168) -> dense tile
//   symmetrize_i32   mirror the computed upper triangle (the reference fills both, :187)
//   fold / export    int32 partial <-> int64 total (Scala Int wraps at 2^31; we do not)
//   synth_fill_f32   counter-based synthetic genotypes (bench / tests only)
// All are plain coalesced streaming kernels; none is on the roofline-critical path.
#include "pcoa_internal.h"

namespace pcoa {
namespace {

// One wave per variant row: lanes stride over the row's carrier list and add 1.0f at
// x[row][sample].  Float atomics keep multiplicity semantics for repeated indices (the reference's
// double loop counts a repeated carrier twice).  Rows are disjoint, so contention only arises
// from genuine repeats.
__global__ __launch_bounds__(256) void densify_csr_kernel(const int32_t* __restrict__ idx,
                                                          const int64_t* __restrict__ offs, int64_t v0,
                                                          int64_t nv, int64_t offs_base,
                                                          float* __restrict__ x, int64_t ld, int32_t n,
                                                          int32_t* __restrict__ err_flag) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nv) return;
  const int64_t b = offs[v0 + row] - offs_base;
  const int64_t e = offs[v0 + row + 1] - offs_base;
  float* xr = x + row * ld;
  for (int64_t p = b + lane; p < e; p += 64) {
    const int32_t c = idx[p];
    if (c < 0 || c >= n) {
      atomicOr(err_flag, 1);
    } else {
      atomicAdd(&xr[c], 1.0f);
    }
  }
}

// s[i][j] = s[j][i] for i > j, 32x32 tiles staged through LDS so both sides stay coalesced.
__global__ __launch_bounds__(256) void symmetrize_i32_kernel(int32_t* __restrict__ s, int32_t n) {
  __shared__ int32_t t[32][33];
  const int ntile = (n + 31) / 32;
  const int bj = blockIdx.x;                               // destination tile column
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  // grid.y strides over the destination tile rows so that the dispatch stays below 2^32 work-items
  for (int bi = blockIdx.y; bi < ntile; bi += gridDim.y) {
    if (bi < bj) continue;  // uniform per workgroup
    // read source tile (rows bj*32.., cols bi*32..) = the upper-triangular one
    for (int r = ty; r < 32; r += 8) {
      const int i = bj * 32 + r, j = bi * 32 + tx;
      t[r][tx] = (i < n && j < n) ? s[(int64_t)i * n + j] : 0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
      const int i = bi * 32 + r, j = bj * 32 + tx;
      if (i < n && j < n && i > j) s[(int64_t)i * n + j] = t[tx][r];
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void fold_kernel(int32_t* __restrict__ s32, int64_t* __restrict__ s64,
                                                   int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (int64_t)gridDim.x * blockDim.x) {
    s64[i] += (int64_t)s32[i];
    s32[i] = 0;
  }
}

__global__ __launch_bounds__(256) void export_kernel(const int32_t* __restrict__ s32,
                                                     const int64_t* __restrict__ s64, int64_t* __restrict__ dst,
                                                     int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (int64_t)gridDim.x * blockDim.x) {
    dst[i] = (int64_t)s32[i] + (s64 ? s64[i] : 0);
  }
}

// ---- Philox4x32-10 (Salmon et al., SC'11), the counter-based generator of the synthetic model ---
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                              uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// One thread per (variant, group of 4 samples): counter = (v_lo, v_hi, i/4, 0), key = seed.
// X[v][i] = 1.0f iff word[i % 4] < thresholds[v_local][pop(i)].
__global__ __launch_bounds__(256) void synth_fill_kernel(uint64_t seed, const uint32_t* __restrict__ thr,
                                                         const int32_t* __restrict__ sample_pop, int32_t n_pops,
                                                         int64_t first_variant, int64_t nv, int32_t n,
                                                         int32_t ngroups, float* __restrict__ x, int64_t ld,
                                                         int vec_ok) {
  const int64_t gid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = nv * ngroups;
  if (gid >= total) return;
  const int64_t vl = gid / ngroups;
  const int32_t g = (int32_t)(gid - vl * ngroups);
  const uint64_t v = (uint64_t)(first_variant + vl);
  uint32_t w[4];
  philox4x32_10((uint32_t)v, (uint32_t)(v >> 32), (uint32_t)g, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), w);
  const uint32_t* tv = thr + vl * n_pops;
  float o[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int i = g * 4 + t;
    o[t] = (i < n && w[t] < tv[sample_pop[i]]) ? 1.0f : 0.0f;
  }
  float* dst = x + vl * ld + (int64_t)g * 4;
  if (vec_ok && g * 4 + 3 < ld) {
    *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
  } else {
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (g * 4 + t < ld) dst[t] = o[t];
  }
}

// The same model written straight into the k-bits operand K1[blk][npad][4 words] (gram_kbits.inl): the fp32 tile -- 400 GB
// for configs[3]'s 100,000 samples x 10^6 variants -- is never written or read back (r06; r05: synth_fill_kernel into a staging
// tile + pack_kbits_kernel, 0.14 + 0.07 s of that job).  One thread = (block of 128 variants, 4 samples): 128 Philox calls,
// whose word t decides sample 4 g + t exactly as synth_fill_kernel's does (same key, same counter, same thresholds), 64
// contiguous bytes out.  thr_s: the block's thresholds [128][n_pops]; rows beyond nv get threshold 0 (no carrier).
__global__ __launch_bounds__(256) void synth_kbits_kernel(uint64_t seed, const uint32_t* __restrict__ thr,
                                                          const int32_t* __restrict__ sample_pop, int32_t n_pops,
                                                          int64_t first_variant, int64_t nv, int32_t n, int32_t npad,
                                                          uint32_t* __restrict__ p) {
  extern __shared__ uint32_t thr_s[];
  const int64_t blk = blockIdx.y;
  const int64_t left = nv - blk * 128;
  const int rows = left >= 128 ? 128 : (left > 0 ? (int)left : 0);
  for (int t = threadIdx.x; t < 128 * n_pops; t += 256) thr_s[t] = (t / n_pops < rows) ? thr[blk * 128 * n_pops + t] : 0u;
  __syncthreads();
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g * 4 >= npad) return;
  int pop[4];
  uint32_t live[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int i = g * 4 + t;
    live[t] = i < n ? 0xffffffffu : 0u;
    pop[t] = i < n ? sample_pop[i] : 0;
  }
  uint32_t w[4][4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    uint32_t acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
#pragma unroll 2
    for (int b = 0; b < 32; ++b) {
      const int vl = c * 32 + b;
      const uint64_t v = (uint64_t)(first_variant + blk * 128 + vl);
      uint32_t x[4];
      philox4x32_10((uint32_t)v, (uint32_t)(v >> 32), (uint32_t)g, 0u, (uint32_t)seed, (uint32_t)(seed >> 32), x);
      const uint32_t* tv = thr_s + vl * n_pops;
      acc0 |= (uint32_t)(x[0] < tv[pop[0]]) << b;
      acc1 |= (uint32_t)(x[1] < tv[pop[1]]) << b;
      acc2 |= (uint32_t)(x[2] < tv[pop[2]]) << b;
      acc3 |= (uint32_t)(x[3] < tv[pop[3]]) << b;
    }
    w[0][c] = acc0 & live[0];
    w[1][c] = acc1 & live[1];
    w[2][c] = acc2 & live[2];
    w[3][c] = acc3 & live[3];
  }
  uint4* dst = reinterpret_cast<uint4*>(p + ((size_t)blk * npad + (size_t)g * 4) * 4);
#pragma unroll
  for (int t = 0; t < 4; ++t) dst[t] = make_uint4(w[t][0], w[t][1], w[t][2], w[t][3]);
}

inline unsigned grid_for(int64_t count, int block, int64_t cap) {
  int64_t g = (count + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

hipError_t launch_densify_csr(const int32_t* idx_dev, const int64_t* offs_dev, int64_t v0, int64_t nv,
                              int64_t offs_base, float* x_dev, int64_t ld, int32_t n, int32_t* err_flag_dev,
                              hipStream_t stream) {
  if (nv <= 0) return hipSuccess;
  const int64_t blocks = (nv + 3) / 4;
  hipLaunchKernelGGL(densify_csr_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, idx_dev, offs_dev, v0, nv,
                     offs_base, x_dev, ld, n, err_flag_dev);
  return hipGetLastError();
}

hipError_t launch_symmetrize_i32(int32_t* s32, int32_t n, hipStream_t stream) {
  const unsigned t = (unsigned)((n + 31) / 32);
  const unsigned ty = t < 2048u ? t : 2048u;  // t * ty * 256 work-items < 2^32 up to N = 262,144
  hipLaunchKernelGGL(symmetrize_i32_kernel, dim3(t, ty), dim3(256), 0, stream, s32, n);
  return hipGetLastError();
}

hipError_t launch_fold_i32_to_i64(int32_t* s32, int64_t* s64, int64_t count, hipStream_t stream) {
  hipLaunchKernelGGL(fold_kernel, dim3(grid_for(count, 256, 8192)), dim3(256), 0, stream, s32, s64, count);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void add_i64_kernel(int64_t* __restrict__ dst, const int64_t* __restrict__ src, int64_t count) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

hipError_t launch_add_i64(int64_t* dst, const int64_t* src, int64_t count, hipStream_t stream) {
  hipLaunchKernelGGL(add_i64_kernel, dim3(grid_for(count, 256, 8192)), dim3(256), 0, stream, dst, src, count);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void add_i32_kernel(int32_t* __restrict__ dst, const int32_t* __restrict__ src, int64_t count) {
  const int64_t quads = count >> 2;
  int4* d4 = reinterpret_cast<int4*>(dst);
  const int4* s4 = reinterpret_cast<const int4*>(src);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < quads; i += (int64_t)gridDim.x * blockDim.x) {
    int4 a = d4[i];
    const int4 b = s4[i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    d4[i] = a;
  }
  if (blockIdx.x == 0 && threadIdx.x < (count & 3)) dst[quads * 4 + threadIdx.x] += src[quads * 4 + threadIdx.x];
}

// dst += src for two finalized int32 partials whose sum is known to stay inside int32 (pcoa_gram_reduce_from, r06)
hipError_t launch_add_i32(int32_t* dst, const int32_t* src, int64_t count, hipStream_t stream) {
  // (hipMalloc / the guard allocator hand out 16-byte aligned buffers; N x N or N x cols from their start)
  hipLaunchKernelGGL(add_i32_kernel, dim3(grid_for((count + 3) / 4, 256, 8192)), dim3(256), 0, stream, dst, src, count);
  return hipGetLastError();
}

// int64 total -> int32 matrix where every entry fits (narrow_s64, pcoa_capi.hip): flag[0] is raised by an entry that does
// not, the 64-bit word at flag + 2 receives max |entry|
__global__ __launch_bounds__(256) void narrow_i64_kernel(const int64_t* __restrict__ s64, int32_t* __restrict__ s32, int64_t count,
                                                         int32_t* __restrict__ flag) {
  __shared__ unsigned long long red[4];
  unsigned long long mx = 0;
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t v = s64[i];
    const unsigned long long a = (unsigned long long)(v < 0 ? -v : v);
    mx = a > mx ? a : mx;
    bad = bad || a >= 0x7fffffffull;
    s32[i] = (int32_t)v;
  }
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = __shfl_down(mx, o, 64);
    mx = other > mx ? other : mx;
  }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  const bool any_bad = __syncthreads_or(bad ? 1 : 0) != 0;
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) mx = red[w] > mx ? red[w] : mx;
    atomicMax(reinterpret_cast<unsigned long long*>(flag + 2), mx);
    if (any_bad) atomicOr(flag, 1);
  }
}

hipError_t launch_narrow_i64_to_i32(const int64_t* s64, int32_t* s32, int64_t count, int32_t* flag, hipStream_t stream) {
  hipLaunchKernelGGL(narrow_i64_kernel, dim3(grid_for(count, 256, 8192)), dim3(256), 0, stream, s64, s32, count, flag);
  return hipGetLastError();
}

// PLINK 1 .bed rows (variant-major, 2 bits per genotype, sample s in bits 2 (s % 4) of byte s / 4: 00 hom A1, 01 missing,
// 10 het, 11 hom A2) -> carrier bitsets (the pcoa_accumulate_bits layout).  hasVariation (VariantsPca.scala:56-60) with A2 the
// reference allele = code 00 or 10; ref_a1: 11 or 10.  One wave per row, one lane per output word = 32 samples = 8 bytes of
// the row.  Rows are ceil(N / 4) bytes apart and so start at any byte: a lane reads the one or two ALIGNED 8-byte units that
// hold its 8 bytes (coalesced; a unit never crosses a page, and the second one is only touched when the row's own bytes
// reach into it) and shifts them together.  (r04 form: eight byte loads per word and a 64-bit division per thread,
// 0.61 ms per 10^6 variants at N = 2504; this one is bound by the 0.94 GB it moves.)
__global__ __launch_bounds__(256) void plink_bed_to_bits_kernel(const uint8_t* __restrict__ bed, int64_t row_bytes, int64_t nv,
                                                                int32_t n, int64_t words, int ref_a1, uint32_t* __restrict__ bits) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= nv) return;
  const uint8_t* row = bed + r * row_bytes;
  for (int64_t w = lane; w < words; w += 64) {
    const int64_t avail = row_bytes - 8 * w;          // bytes the row still has at this word (>= 1: 8 (words - 1) < N / 4)
    uint64_t x = 0;
    if (avail > 0) {
      const int need = avail < 8 ? (int)avail : 8;
      const uintptr_t a = reinterpret_cast<uintptr_t>(row + 8 * w);
      const int sh = (int)(a & 7);
      const uint64_t* q = reinterpret_cast<const uint64_t*>(a - sh);
      x = q[0] >> (8 * sh);
      if (sh + need > 8) x |= q[1] << (8 * (8 - sh));   // (sh > 0 here)
      if (need < 8) x &= (1ull << (8 * need)) - 1ull;   // what follows the row is not its own
    }
    const uint64_t E = 0x5555555555555555ull;
    const uint64_t lo = x & E, hi = (x >> 1) & E;
    uint64_t m = (hi & ~lo) | (ref_a1 ? (hi & lo) : (~hi & ~lo & E));  // carrier flags in the even bit positions
    m = (m | (m >> 1)) & 0x3333333333333333ull;                            // ... squeezed into the low 32 bits
    m = (m | (m >> 2)) & 0x0f0f0f0f0f0f0f0full;
    m = (m | (m >> 4)) & 0x00ff00ff00ff00ffull;
    m = (m | (m >> 8)) & 0x0000ffff0000ffffull;
    m = (m | (m >> 16)) & 0x00000000ffffffffull;
    const int64_t first = 32 * w;
    uint32_t keep = 0xffffffffu;
    if (first >= n) keep = 0u;
    else if (first + 32 > n) keep = (1u << (n - (int32_t)first)) - 1u;
    bits[r * words + w] = (uint32_t)m & keep;
  }
}

hipError_t launch_plink_bed_to_bits(const uint8_t* bed, int64_t row_bytes, int64_t nv, int32_t n, int64_t words, int ref_a1,
                                    uint32_t* bits, hipStream_t stream) {
  if (nv <= 0) return hipSuccess;
  const int64_t blocks = (nv + 3) / 4;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(plink_bed_to_bits_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, bed, row_bytes, nv, n, words, ref_a1,
                     bits);
  return hipGetLastError();
}

hipError_t launch_export_i64(const int32_t* s32, const int64_t* s64_or_null, int64_t* dst, int64_t count,
                             hipStream_t stream) {
  hipLaunchKernelGGL(export_kernel, dim3(grid_for(count, 256, 8192)), dim3(256), 0, stream, s32, s64_or_null,
                     dst, count);
  return hipGetLastError();
}

hipError_t launch_synth_fill_f32(uint64_t seed, const uint32_t* thresholds_dev, const int32_t* sample_pop_dev,
                                 int32_t n_pops, int64_t first_variant, int64_t nv, int32_t n, float* x_dev,
                                 int64_t ld, hipStream_t stream) {
  if (nv <= 0) return hipSuccess;
  const int32_t ngroups = (int32_t)((ld + 3) / 4);  // also zero-fills the padding columns [n, ld)
  const int64_t total = nv * ngroups;
  const int vec_ok = ((ld & 3) == 0) && ((reinterpret_cast<uintptr_t>(x_dev) & 15) == 0);
  const int64_t blocks = (total + 255) / 256;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  hipLaunchKernelGGL(synth_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, seed, thresholds_dev,
                     sample_pop_dev, n_pops, first_variant, nv, n, ngroups, x_dev, ld, vec_ok);
  return hipGetLastError();
}

hipError_t launch_synth_kbits(uint64_t seed, const uint32_t* thresholds_dev, const int32_t* sample_pop_dev, int32_t n_pops,
                              int64_t first_variant, int64_t nv, int32_t n, int32_t npad, int8_t* p, int64_t nblk_out,
                              hipStream_t stream) {
  if (nblk_out <= 0) return hipSuccess;
  if (nblk_out > 65535 || n_pops <= 0 || n_pops > 64) return hipErrorInvalidValue;
  const unsigned gx = (unsigned)((npad / 4 + 255) / 256);
  hipLaunchKernelGGL(synth_kbits_kernel, dim3(gx, (unsigned)nblk_out), dim3(256), (size_t)(128 * n_pops) * sizeof(uint32_t), stream,
                     seed, thresholds_dev, sample_pop_dev, n_pops, first_variant, nv, n, npad, reinterpret_cast<uint32_t*>(p));
  return hipGetLastError();
}

namespace {
// one wave that does nothing for `ticks` of the 100-MHz wall clock
__global__ void delay_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
}  // namespace

hipError_t launch_delay_us(hipStream_t stream, int microseconds) {
  hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, stream, (long long)microseconds * 100);
  return hipGetLastError();
}

}  // namespace pcoa
