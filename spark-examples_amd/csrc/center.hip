// center.hip -- computePca part 1+2 (reference VariantsPca.scala:199-223; variants_pca.py:84-121):
//
//   rowSums(i)  = sum_j S(i,j)                               (:206)
//   nonZeroRows = #{ rowSums > 0 }                           (:207)
//   matrixMean  = (sum_i rowSums(i)) / N / N                 (:210-211)  two divisions, in that order
//   B(i,j)      = ((S(i,j) - rowSums(i)/N) - rowSums(j)/N) + matrixMean   (:216-221)
//
// Row sums are accumulated as int64 (exact); the reference folds doubles left to right, which is
// also exact while the sums stay below 2^53, so both give the same double.  The centring expression
// is evaluated in the reference's operation order with IEEE fp64 +,-,/ and no fused multiply-add,
// so B is bit-identical to the JVM's result.  HBM-bound streaming: reads 4 (or 12) B, writes 8 B
// per entry.
#include <type_traits>
#include "pcoa_internal.h"

namespace pcoa {
namespace {

__device__ __forceinline__ int64_t wave_sum_i64(int64_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// one workgroup (256 threads) per row
__global__ __launch_bounds__(256) void row_sums_kernel(const int32_t* __restrict__ s32,
                                                       const int64_t* __restrict__ s64, int32_t n,
                                                       double* __restrict__ row_sums,
                                                       int64_t* __restrict__ row_sums_i64) {
  __shared__ int64_t part[4];
  const int i = blockIdx.x;
  const int64_t base = (int64_t)i * n;
  int64_t acc = 0;
  for (int j = threadIdx.x; j < n; j += 256) acc += (int64_t)s32[base + j] + (s64 ? s64[base + j] : 0);
  acc = wave_sum_i64(acc);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int64_t t = part[0] + part[1] + part[2] + part[3];
    row_sums_i64[i] = t;
    row_sums[i] = (double)t;
  }
}

// single workgroup: matrix sum, mean, non-zero rows
__global__ __launch_bounds__(256) void stats_kernel(const int64_t* __restrict__ row_sums_i64, int32_t n,
                                                    double* __restrict__ stats, int32_t* __restrict__ nz) {
  __shared__ int64_t part[4];
  __shared__ int64_t cnt[4];
  int64_t acc = 0, c = 0;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int64_t r = row_sums_i64[i];
    acc += r;
    c += (r > 0) ? 1 : 0;
  }
  acc = wave_sum_i64(acc);
  c = wave_sum_i64(c);
  if ((threadIdx.x & 63) == 0) {
    part[threadIdx.x >> 6] = acc;
    cnt[threadIdx.x >> 6] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma clang fp contract(off)
    const double msum = (double)(part[0] + part[1] + part[2] + part[3]);
    const double rc = (double)n;
    stats[0] = msum;
    stats[1] = msum / rc / rc;
    nz[0] = (int32_t)(cnt[0] + cnt[1] + cnt[2] + cnt[3]);
  }
}

__global__ __launch_bounds__(256) void center_kernel(const int32_t* __restrict__ s32,
                                                     const int64_t* __restrict__ s64, int32_t n,
                                                     const double* __restrict__ row_sums,
                                                     const double* __restrict__ stats, double* __restrict__ b) {
  // One workgroup per row, striding over the columns: the dispatch stays at N x 256 work-items.
  // (A block per (row, 256-column chunk) is 1.0e10 work-items at N = 100,000 -- more than the 32-bit
  // work-item count of a dispatch holds; it silently centred only part of the matrix.)
#pragma clang fp contract(off)
  const int i = blockIdx.x;
  const double rc = (double)n;
  const double row_mean = row_sums[i] / rc;
  const double mmean = stats[1];
  for (int j = threadIdx.x; j < n; j += 256) {
    const double col_mean = row_sums[j] / rc;
    const int64_t idx = (int64_t)i * n + j;
    const double data = (double)((int64_t)s32[idx] + (s64 ? s64[idx] : 0));
    double t = data - row_mean;
    t = t - col_mean;
    t = t + mmean;
    b[idx] = t;
  }
}

__global__ __launch_bounds__(256) void col_means_kernel(const double* __restrict__ row_sums, int32_t n,
                                                        double* __restrict__ cm) {
#pragma clang fp contract(off)
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j < n) cm[j] = row_sums[j] / (double)n;
}

// ---- strip owner: S is [n][cols] row-major.  Both reductions run over the ROWS for every column: a
// workgroup takes 256 columns x a band of 256 rows (coalesced along the columns), writes one partial per column, and a
// second pass adds the bands in a fixed order -- deterministic, no floating-point atomics.
constexpr int STRIP_BAND = 256;

// r05: a workgroup takes 1024 columns x a band of 256 rows.  Each of its four waves walks ALL rows of the band over its own 256
// columns, a lane owning the four columns lane + 64 q: four independent 4-byte loads per row (each wave-level load is 256
// contiguous bytes -- any `cols`, no alignment rule), rows prefetched four deep (16 loads per lane in flight), the band's
// v_i / rowMean_i staged in LDS once.  The r04 form (one column per thread, one load in flight per iteration, v_i and
// rowMean_i re-read per row) streamed the strip at a fraction of the HBM rate; the additions per column happen in the same
// order as before (rows of a band in order, bands in order): results are bit-identical.
constexpr int STRIP_NB = 4;   // row buffers

template <bool MATVEC, bool HAS64>
__global__ __launch_bounds__(256) void strip_band_kernel(const int32_t* __restrict__ s32, const int64_t* __restrict__ s64,
                                                         int32_t n, int32_t col0, int32_t cols, const double* __restrict__ v,
                                                         const double* __restrict__ means, double mmean,
                                                         double* __restrict__ partial /* [bands][cols] */) {
#pragma clang fp contract(off)
  __shared__ double vs[STRIP_BAND], ms[STRIP_BAND];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int band = blockIdx.y;
  const int i0 = band * STRIP_BAND, i1 = min(n, i0 + STRIP_BAND);
  const int rows = i1 - i0;
  if constexpr (MATVEC) {
    for (int r = threadIdx.x; r < STRIP_BAND; r += 256) {
      vs[r] = r < rows ? v[i0 + r] : 0.0;
      ms[r] = r < rows ? means[i0 + r] : 0.0;
    }
    __syncthreads();
  }
  const int jbase = (blockIdx.x * 4 + wave) * 256 + lane;   // this lane's columns: jbase + 64 q
  if (jbase - lane >= cols) return;                         // (wave-uniform: the whole wave lies beyond the strip)
  // FULL: all 256 columns of the wave lie inside the strip (wave-uniform): no masks in the row loop
  auto run = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    bool ok[4];
    double mj[4], acc[4];
    int64_t iacc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int j = jbase + 64 * q;
      ok[q] = FULL || j < cols;
      mj[q] = (MATVEC && ok[q]) ? means[col0 + j] : 0.0;   // the strip's column j is row j of the symmetric matrix
      acc[q] = 0.0;
      iacc[q] = 0;
    }
    const int64_t base = (int64_t)i0 * cols + jbase;
    // (the buffers hold what was LOADED -- a value converted at load time would be waited for at once)
    struct Row { int32_t lo[4]; int64_t hi[HAS64 ? 4 : 1]; };
    auto load_row = [&](int r, Row& b) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t idx = base + (int64_t)r * cols + 64 * q;
        b.lo[q] = (FULL || ok[q]) ? s32[idx] : 0;
        if constexpr (HAS64) b.hi[q] = (FULL || ok[q]) ? s64[idx] : 0;
      }
    };
    auto use_row = [&](int r, const Row& b) {
      if constexpr (MATVEC) {
        const double vi = vs[r], mi = ms[r];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const double data = HAS64 ? (double)((int64_t)b.lo[q] + b.hi[HAS64 ? q : 0]) : (double)b.lo[q];
          double t = data - mj[q];
          t = t - mi;
          t = t + mmean;
          acc[q] += t * vi;
        }
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) iacc[q] += (int64_t)b.lo[q] + (HAS64 ? b.hi[HAS64 ? q : 0] : 0);
      }
    };
    Row buf[STRIP_NB];
#pragma unroll
    for (int b = 0; b < STRIP_NB; ++b)
      if (b < rows) load_row(b, buf[b]);
    int k = 0;
    for (; k + 2 * STRIP_NB <= rows; k += STRIP_NB) {
#pragma unroll
      for (int b = 0; b < STRIP_NB; ++b) {
        use_row(k + b, buf[b]);
        load_row(k + b + STRIP_NB, buf[b]);
      }
    }
#pragma unroll
    for (int b = 0; b < STRIP_NB; ++b)
      if (k + b < rows) {
        use_row(k + b, buf[b]);
        if (k + b + STRIP_NB < rows) load_row(k + b + STRIP_NB, buf[b]);
      }
    k += STRIP_NB;
#pragma unroll
    for (int b = 0; b < STRIP_NB; ++b)
      if (k + b < rows) use_row(k + b, buf[b]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (!ok[q]) continue;
      const int64_t o = (int64_t)band * cols + jbase + 64 * q;
      if constexpr (MATVEC) partial[o] = acc[q];
      else reinterpret_cast<int64_t*>(partial)[o] = iacc[q];
    }
  };
  if (jbase - lane + 256 <= cols) run(std::true_type{});
  else run(std::false_type{});
}

template <bool MATVEC>
__global__ __launch_bounds__(256) void strip_finish_kernel(const double* __restrict__ partial, int32_t cols, int32_t bands,
                                                           double* __restrict__ out) {
  const int jj = blockIdx.x * 256 + threadIdx.x;
  if (jj >= cols) return;
  if constexpr (MATVEC) {
    double acc = 0.0;
    for (int b = 0; b < bands; ++b) acc += partial[(int64_t)b * cols + jj];
    out[jj] = acc;
  } else {
    int64_t acc = 0;
    for (int b = 0; b < bands; ++b) acc += reinterpret_cast<const int64_t*>(partial)[(int64_t)b * cols + jj];
    out[jj] = (double)acc;
  }
}

}  // namespace

int64_t strip_ws_doubles(int32_t n, int32_t cols) {
  const int64_t bands = ((int64_t)n + STRIP_BAND - 1) / STRIP_BAND;
  return (bands + 1) * (int64_t)cols;   // [0, cols): the result; behind it the per-band partials
}

hipError_t launch_strip_col_sums(const int32_t* s32, const int64_t* s64_or_null, int32_t n, int32_t cols, double* ws,
                                 hipStream_t stream) {
  const int bands = (n + STRIP_BAND - 1) / STRIP_BAND;
  const dim3 grid((unsigned)((cols + 1023) / 1024), (unsigned)bands);
  if (s64_or_null)
    hipLaunchKernelGGL((strip_band_kernel<false, true>), grid, dim3(256), 0, stream, s32, s64_or_null, n, 0, cols, nullptr, nullptr,
                       0.0, ws + cols);
  else
    hipLaunchKernelGGL((strip_band_kernel<false, false>), grid, dim3(256), 0, stream, s32, s64_or_null, n, 0, cols, nullptr, nullptr,
                       0.0, ws + cols);
  hipLaunchKernelGGL(strip_finish_kernel<false>, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, stream, ws + cols, cols, bands, ws);
  return hipGetLastError();
}

hipError_t launch_strip_matvec(const int32_t* s32, const int64_t* s64_or_null, int32_t n, int32_t col0, int32_t cols,
                               const double* v, const double* means, double matrix_mean, double* ws, hipStream_t stream) {
  const int bands = (n + STRIP_BAND - 1) / STRIP_BAND;
  const dim3 grid((unsigned)((cols + 1023) / 1024), (unsigned)bands);
  if (s64_or_null)
    hipLaunchKernelGGL((strip_band_kernel<true, true>), grid, dim3(256), 0, stream, s32, s64_or_null, n, col0, cols, v, means,
                       matrix_mean, ws + cols);
  else
    hipLaunchKernelGGL((strip_band_kernel<true, false>), grid, dim3(256), 0, stream, s32, s64_or_null, n, col0, cols, v, means,
                       matrix_mean, ws + cols);
  hipLaunchKernelGGL(strip_finish_kernel<true>, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, stream, ws + cols, cols, bands, ws);
  return hipGetLastError();
}

hipError_t launch_col_means(const double* row_sums, int32_t n, double* cm, hipStream_t stream) {
  hipLaunchKernelGGL(col_means_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, row_sums, n, cm);
  return hipGetLastError();
}

hipError_t launch_center(const int32_t* s32, const int64_t* s64_or_null, int32_t n, double* row_sums,
                         double* stats, int32_t* nz, double* b, hipStream_t stream, bool row_sums_done) {
  // row_sums_i64 lives right behind stats[0..1] (the caller allocates stats as 2 + n doubles)
  int64_t* rs_i64 = reinterpret_cast<int64_t*>(stats + 2);
  if (!row_sums_done)   // (large N: launch_row_sums_sym has filled row_sums / rs_i64 from the upper triangle)
    hipLaunchKernelGGL(row_sums_kernel, dim3((unsigned)n), dim3(256), 0, stream, s32, s64_or_null, n, row_sums,
                       rs_i64);
  hipLaunchKernelGGL(stats_kernel, dim3(1), dim3(256), 0, stream, rs_i64, n, stats, nz);
  if (b) {
    hipLaunchKernelGGL(center_kernel, dim3((unsigned)n), dim3(256), 0, stream, s32, s64_or_null, n, row_sums, stats,
                       b);
  }
  return hipGetLastError();
}

}  // namespace pcoa
