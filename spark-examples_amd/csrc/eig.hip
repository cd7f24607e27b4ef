// eig.hip -- symmetric eigensolver for the top principal coordinates of the centred matrix B.
//
// Replaces computePca part 3 (reference VariantsPca.scala:224-227): MLlib 1.6.1
// RowMatrix.computePrincipalComponents = covariance + Breeze svd (LAPACK dgesdd) on one driver
// thread.  For symmetric B the principal components are the eigenvectors of B with the largest
// |lambda| (Cov = B^T B/(N-1) - N/(N-1) mu mu^T has the same eigenvectors, singular values
// lambda^2/(N-1)), so B is eigendecomposed directly, all in fp64:
//
//   K3  Householder tridiagonalisation  B = Q T Q^T              (tridiag_hw_kernel + tridiag_update_kernel)
//   K4  top-k eigenvalues of T by Sturm-count multisection        (bisect_kernel)
//       eigenvectors of T by inverse iteration (pivoted LU)       (invit_kernel)
//   K5  back-transform z <- Q z, normalise, sign-normalise        (backtransform_kernel)
//
// K3 keeps the full symmetric matrix (row i == column i) so that every access is a coalesced row
// access, and DEFERS each rank-2 update: step k's kernel applies step k-1's update
//     A22 <- A22 - v w^T - w v^T
// to the trailing rows while it computes the next matrix-vector product q = A22 v_k in the same
// pass (one read + one write of the trailing block per column instead of two passes).  One wave
// owns one row, so q_i is produced by one wave in a fixed order: bit-reproducible, no atomics.
// The tiny serial part of each step (norms, tau, w) runs in a single 1024-thread workgroup.
//
// K4 deviates from the "implicit QR" wording of BASELINE.json on purpose: implicit QL/QR is a
// serial chain of O(N^2) dependent rotations (one lane busy), whereas Sturm counts for 256 shifts
// at once fill a workgroup and only the k wanted eigenvalues are ever computed.  Same eigenvalues
// to machine precision; see DESIGN_HISTORY.md 4.4.
//
// Bounds: K3 is HBM/L2-bandwidth- and launch-latency-bound (BLAS-2), K4/K5 are latency-bound;
// the eigensolver is reported as wall-clock, not against a roofline.
#include <cfloat>
#include <type_traits>
#include <algorithm>
#include <cmath>

#include "pcoa_internal.h"

namespace pcoa {
namespace {

constexpr int HW_T = 1024;  // threads of the single-workgroup kernels

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;  // valid in lane 0
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_down(v, o, 64));
  return v;
}

// Block-wide reductions, result broadcast to every thread.  red: >= 17 doubles of LDS.
// Every thread of the block must call.  OP: 0 sum, 1 max, 2 min.
template <int OP>
__device__ __forceinline__ double block_reduce(double v, double* red) {
  v = (OP == 0) ? wave_sum(v) : (OP == 1) ? wave_max(v) : wave_min(v);
  const int w = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  __syncthreads();  // red may still be read from a previous reduction
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  if (threadIdx.x < 64) {
    const double ident = (OP == 0) ? 0.0 : (OP == 1) ? -DBL_MAX : DBL_MAX;
    double t = ((int)threadIdx.x < nw) ? red[threadIdx.x] : ident;
    t = (OP == 0) ? wave_sum(t) : (OP == 1) ? wave_max(t) : wave_min(t);
    if (threadIdx.x == 0) red[16] = t;
  }
  __syncthreads();
  return red[16];
}

// ------------------------------------------------------------------------------------------------
// K3a: serial part of Householder step k (single workgroup).
//   * finishes step k-1:  w = tau (q - 1/2 tau (q.v) v)            (v = v_{k-1}, stored in row k-1)
//   * forms column k of the updated matrix without touching the rest:
//         x_j = A[k][j] - v_k' w_j - w_k' v_j   (j > k),   d[k] = A[k][k] - 2 v_k' w_k'
//   * generates the reflector (LAPACK dlarfg): beta = -sign(alpha) ||x||, tau = (beta - alpha)/beta,
//     v = x / (alpha - beta), v[k+1] = 1; stores v in row k (dead from now on), e[k] = beta.
//   For k == n-2 only d[n-2], e[n-2], d[n-1] remain.
__global__ __launch_bounds__(HW_T) void tridiag_hw_kernel(double* __restrict__ a, int n, int k,
                                                          double* __restrict__ d, double* __restrict__ e,
                                                          double* __restrict__ tau, const double* __restrict__ q,
                                                          double* __restrict__ w) {
  __shared__ double red[24];
  const int tid = threadIdx.x;
  const int nt = blockDim.x;  // 256 for short rows (cheaper barriers), 1024 for long ones
  double* rowk = a + (int64_t)k * n;
  const bool pending = k > 0;
  const double* vprev = a + (int64_t)(pending ? k - 1 : 0) * n;
  double wk = 0.0, vpk = 0.0;
  if (pending) {
    const double tp = tau[k - 1];
    double part = 0.0;
    for (int j = k + tid; j < n; j += nt) part += q[j] * vprev[j];
    const double dot = block_reduce<0>(part, red);
    const double c = 0.5 * tp * tp * dot;
    for (int j = k + tid; j < n; j += nt) {
      const double wj = tp * q[j] - c * vprev[j];
      w[j] = wj;
      if (j == k) red[20] = wj;
    }
    __syncthreads();
    wk = red[20];
    vpk = vprev[k];
  }
  if (tid == 0) d[k] = pending ? rowk[k] - 2.0 * vpk * wk : rowk[k];
  if (k >= n - 1) return;  // n == 1

  double part = 0.0;
  for (int j = k + tid; j < n; j += nt) {
    if (j == k) continue;
    double xj = rowk[j];
    if (pending) xj -= vpk * w[j] + wk * vprev[j];
    rowk[j] = xj;
    if (j == k + 1) red[21] = xj; else part += xj * xj;
    if (k == n - 2) {  // j == n-1: last off-diagonal and last diagonal entry
      e[k] = xj;
      const double ann = a[(int64_t)(n - 1) * n + (n - 1)];
      d[n - 1] = pending ? ann - 2.0 * vprev[n - 1] * w[n - 1] : ann;
    }
  }
  if (k == n - 2) return;

  const double xnorm2 = block_reduce<0>(part, red);  // its barriers also publish red[21]
  const double alpha = red[21];
  double beta, t, scale;
  if (xnorm2 == 0.0) {
    beta = alpha; t = 0.0; scale = 0.0;
  } else {
    beta = -copysign(sqrt(alpha * alpha + xnorm2), alpha);
    t = (beta - alpha) / beta;
    scale = 1.0 / (alpha - beta);
  }
  for (int j = k + tid; j < n; j += nt) {
    if (j == k) continue;
    rowk[j] = (j == k + 1) ? 1.0 : rowk[j] * scale;
  }
  if (tid == 0) {
    e[k] = beta;
    tau[k] = t;
  }
}

// K3b: parallel part of step k -- one wave per trailing row i in (k, n):
//   A[i][j] -= v'_i w'_j + w'_i v'_j   (pending update of step k-1, j > k)      read + write
//   q[i]     = sum_j A[i][j] * v_j     (v = v_k, row k)                          same pass
__global__ __launch_bounds__(256) void tridiag_update_kernel(double* __restrict__ a, int n, int k,
                                                             const double* __restrict__ w,
                                                             double* __restrict__ q) {
  const int lane = threadIdx.x & 63;
  const int i = k + 1 + blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const bool pending = k > 0;
  const double* vk = a + (int64_t)k * n;
  const double* vprev = a + (int64_t)(pending ? k - 1 : 0) * n;
  double* row = a + (int64_t)i * n;
  double vpi = 0.0, wpi = 0.0;
  if (pending) {
    vpi = vprev[i];
    wpi = w[i];
  }
  double acc = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
  int j = ((k + 1) & ~63) + lane;  // 512-B aligned start; the first chunk is masked at j <= k
  if (pending) {
    if (j > k && j < n) {
      const double aij = row[j] - (vpi * w[j] + wpi * vprev[j]);
      row[j] = aij;
      acc += aij * vk[j];
    }
    j += 64;
    // four independent 512-B row segments in flight per wave (memory-level parallelism: the pass is
    // bandwidth/latency-bound, ~10 waves per CU)
    for (; j + 192 < n; j += 256) {
      const double r0 = row[j], r1 = row[j + 64], r2 = row[j + 128], r3 = row[j + 192];
      const double a0 = r0 - (vpi * w[j] + wpi * vprev[j]);
      const double a1 = r1 - (vpi * w[j + 64] + wpi * vprev[j + 64]);
      const double a2 = r2 - (vpi * w[j + 128] + wpi * vprev[j + 128]);
      const double a3 = r3 - (vpi * w[j + 192] + wpi * vprev[j + 192]);
      row[j] = a0; row[j + 64] = a1; row[j + 128] = a2; row[j + 192] = a3;
      acc += a0 * vk[j];
      acc1 += a1 * vk[j + 64];
      acc2 += a2 * vk[j + 128];
      acc3 += a3 * vk[j + 192];
    }
    for (; j < n; j += 64) {
      const double aij = row[j] - (vpi * w[j] + wpi * vprev[j]);
      row[j] = aij;
      acc += aij * vk[j];
    }
  } else {
    if (j > k && j < n) acc += row[j] * vk[j];
    j += 64;
    for (; j + 192 < n; j += 256) {
      acc += row[j] * vk[j];
      acc1 += row[j + 64] * vk[j + 64];
      acc2 += row[j + 128] * vk[j + 128];
      acc3 += row[j + 192] * vk[j + 192];
    }
    for (; j < n; j += 64) acc += row[j] * vk[j];
  }
  acc = wave_sum((acc + acc1) + (acc2 + acc3));
  if (lane == 0) q[i] = acc;
}

// ------------------------------------------------------------------------------------------------
// K3 fused: ONE launch per Householder step (the two-kernel form above pays two kernel boundaries per column and is
// latency-bound: 17 us per column at N = 2504).  Every workgroup first recomputes the short serial part of step k for
// itself, into LDS -- w_{k-1} from the raw mat-vec q_{k-1}, column k of the updated matrix, the reflector v_k -- and
// then does its share of the parallel part with v_{k-1}, w_{k-1}, v_k served from LDS: the rank-2 update of step k-1 on
// its rows and q_k = A v_k in the same pass.  Nothing a workgroup needs is produced by another workgroup of the same
// launch: q and the current reflector are double-buffered across launches (q[k & 1], vbuf[k & 1]); workgroup 0 is the
// one that publishes d, e, tau, v_k, and moves v_{k-1} into row k-1 of A (dead for everybody by then) for the
// back-transform.  Rows are dealt to workgroups by ABSOLUTE index (row i -> workgroup i mod gridDim, so XCD i mod 8
// whatever k is): a row's 20 KB stay in the same XCD's L2 from step to step instead of migrating as the trailing block
// shrinks.  Same formulas as tridiag_hw_kernel / tridiag_update_kernel; only the order of the reductions differs.
// LDS: 3 (n - k) doubles (v_{k-1}, w_{k-1}, v_k), i.e. n <= 2,730 without raising the dynamic-LDS limit, n <= 6,800 with.
constexpr int FT = 512;  // threads of a fused-step workgroup: 8 waves share one copy of the three vectors
__global__ __launch_bounds__(FT) void tridiag_fused_kernel(double* __restrict__ a, int n, int k, double* __restrict__ d,
                                                            double* __restrict__ e, double* __restrict__ tau,
                                                            const double* __restrict__ q_in, double* __restrict__ q_out,
                                                            const double* __restrict__ v_in, double* __restrict__ v_out) {
  extern __shared__ __attribute__((aligned(16))) double fl[];
  __shared__ double red[24];
  const int tid = threadIdx.x;
  const int len = n - k;            // active indices k .. n-1 <-> 0 .. len-1
  double* vp = fl;                  // v_{k-1}
  double* wl = fl + len;            // w_{k-1}
  double* vk = fl + 2 * len;        // v_k
  const bool pending = k > 0;
  double* rowk = a + (int64_t)k * n;
  // ---- serial part of step k, redundantly per workgroup
  double wk = 0.0, vpk = 0.0;
  if (pending) {
    const double tp = tau[k - 1];
    double part = 0.0;
    for (int j = tid; j < len; j += FT) {
      const double vj = v_in[k + j];
      vp[j] = vj;
      part += q_in[k + j] * vj;
    }
    const double dot = block_reduce<0>(part, red);
    const double c = 0.5 * tp * tp * dot;
    for (int j = tid; j < len; j += FT) wl[j] = tp * q_in[k + j] - c * vp[j];
    __syncthreads();
    wk = wl[0];
    vpk = vp[0];
    if (blockIdx.x == 0) {  // the reflector of step k-1 moves to its final place: row k-1, columns k .. n-1
      double* rowp = a + (int64_t)(k - 1) * n;
      for (int j = tid; j < len; j += FT) rowp[k + j] = vp[j];
    }
  }
  double part = 0.0, alpha = 0.0;
  for (int j = 1 + tid; j < len; j += FT) {
    double xj = rowk[k + j];
    if (pending) xj -= vpk * wl[j] + wk * vp[j];
    vk[j] = xj;
    if (j == 1) alpha = xj; else part += xj * xj;
  }
  const double xnorm2 = block_reduce<0>(part, red);
  if (tid == 0) red[21] = alpha;    // thread 0 owns j = 1
  __syncthreads();
  alpha = red[21];
  double beta, t, scale;
  if (xnorm2 == 0.0) {
    beta = alpha; t = 0.0; scale = 0.0;
  } else {
    beta = -copysign(sqrt(alpha * alpha + xnorm2), alpha);
    t = (beta - alpha) / beta;
    scale = 1.0 / (alpha - beta);
  }
  for (int j = 1 + tid; j < len; j += FT) vk[j] = (j == 1) ? 1.0 : vk[j] * scale;
  if (tid == 0) vk[0] = 0.0;
  __syncthreads();
  if (blockIdx.x == 0) {
    for (int j = tid; j < len; j += FT) v_out[k + j] = vk[j];
    if (tid == 0) {
      d[k] = pending ? rowk[k] - 2.0 * vpk * wk : rowk[k];
      e[k] = beta;
      tau[k] = t;
    }
  }
  // ---- parallel part: rows i > k with i mod gridDim == blockIdx, one wave per row at a time
  const int lane = tid & 63, wave = tid >> 6;
  const int nb = (int)gridDim.x;
  int first = (k + 1) + ((int)blockIdx.x - (k + 1) % nb + nb) % nb;   // smallest i >= k+1 with i mod nb == blockIdx
  for (int i = first + wave * nb; i < n; i += (FT / 64) * nb) {
    double* row = a + (int64_t)i * n + k;   // row[j] <-> column k + j
    const double vpi = pending ? vp[i - k] : 0.0, wpi = pending ? wl[i - k] : 0.0;
    double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
    int j = 1 + lane;
    if (pending) {
      for (; j + 192 < len; j += 256) {
        const double r0 = row[j], r1 = row[j + 64], r2 = row[j + 128], r3 = row[j + 192];
        const double a0 = r0 - (vpi * wl[j] + wpi * vp[j]);
        const double a1 = r1 - (vpi * wl[j + 64] + wpi * vp[j + 64]);
        const double a2 = r2 - (vpi * wl[j + 128] + wpi * vp[j + 128]);
        const double a3 = r3 - (vpi * wl[j + 192] + wpi * vp[j + 192]);
        row[j] = a0; row[j + 64] = a1; row[j + 128] = a2; row[j + 192] = a3;
        acc0 += a0 * vk[j];
        acc1 += a1 * vk[j + 64];
        acc2 += a2 * vk[j + 128];
        acc3 += a3 * vk[j + 192];
      }
      for (; j < len; j += 64) {
        const double aij = row[j] - (vpi * wl[j] + wpi * vp[j]);
        row[j] = aij;
        acc0 += aij * vk[j];
      }
    } else {
      for (; j + 192 < len; j += 256) {
        acc0 += row[j] * vk[j];
        acc1 += row[j + 64] * vk[j + 64];
        acc2 += row[j + 128] * vk[j + 128];
        acc3 += row[j + 192] * vk[j + 192];
      }
      for (; j < len; j += 64) acc0 += row[j] * vk[j];
    }
    const double acc = wave_sum((acc0 + acc1) + (acc2 + acc3));
    if (lane == 0) q_out[i] = acc;
  }
}

// ------------------------------------------------------------------------------------------------
// K4a: eigenvalue number idx[b] (ascending, 0-based) of the symmetric tridiagonal T(d, e) by
// multisection on the Sturm count (LAPACK dstebz's recurrence, 256 shifts per round).
//   count(x) = #{ i : q_i < 0 },  q_0 = d_0 - x,  q_i = d_i - x - e_{i-1}^2 / q_{i-1}
__global__ __launch_bounds__(256) void bisect_kernel(const double* __restrict__ d, const double* __restrict__ e,
                                                     int n, const int* __restrict__ idx,
                                                     double* __restrict__ lam_out, int use_lds) {
  extern __shared__ __attribute__((aligned(16))) double sm[];
  double* red = sm;             // 32 doubles
  double* ds = sm + 32;         // n
  double* e2s = sm + 32 + n;    // n
  const int tid = threadIdx.x;
  const int m = idx[blockIdx.x];

  // Gershgorin bounds, pivmin
  double gl = DBL_MAX, gu = -DBL_MAX, e2max = 0.0;
  for (int i = tid; i < n; i += 256) {
    const double el = (i > 0) ? fabs(e[i - 1]) : 0.0;
    const double er = (i < n - 1) ? fabs(e[i]) : 0.0;
    gl = fmin(gl, d[i] - el - er);
    gu = fmax(gu, d[i] + el + er);
    e2max = fmax(e2max, er * er);
    if (use_lds) {
      ds[i] = d[i];
      e2s[i] = er * er;  // e2s[i] = e[i]^2 (0 for i = n-1)
    }
  }
  gl = block_reduce<2>(gl, red);
  gu = block_reduce<1>(gu, red);
  e2max = block_reduce<1>(e2max, red);
  const double eps = DBL_EPSILON;
  const double pivmin = DBL_MIN * fmax(1.0, e2max);
  const double tnorm = fmax(fabs(gl), fabs(gu));
  gl -= 2.1 * tnorm * eps * n + 2.1 * pivmin;
  gu += 2.1 * tnorm * eps * n + 2.1 * pivmin;

  double lo = gl, hi = gu;
  for (int it = 0; it < 64; ++it) {
    const double x = lo + (hi - lo) * ((double)(tid + 1) / 257.0);
    int c;
    {
      double qv = (use_lds ? ds[0] : d[0]) - x;
      if (fabs(qv) < pivmin) qv = -pivmin;
      c = (qv < 0.0) ? 1 : 0;
      if (use_lds) {
        for (int i = 1; i < n; ++i) {
          qv = ds[i] - x - e2s[i - 1] / qv;
          if (fabs(qv) < pivmin) qv = -pivmin;
          c += (qv < 0.0) ? 1 : 0;
        }
      } else {
        for (int i = 1; i < n; ++i) {
          const double ei = e[i - 1];
          qv = d[i] - x - ei * ei / qv;
          if (fabs(qv) < pivmin) qv = -pivmin;
          c += (qv < 0.0) ? 1 : 0;
        }
      }
    }
    const double cand_lo = (c <= m) ? x : lo;
    const double cand_hi = (c > m) ? x : hi;
    const double nlo = block_reduce<1>(cand_lo, red);
    const double nhi = block_reduce<2>(cand_hi, red);
    if (!(nlo > lo || nhi < hi) || !(nlo < nhi)) break;
    lo = nlo;
    hi = nhi;
    if (hi - lo <= 2.0 * eps * fmax(fabs(lo), fabs(hi)) + 2.0 * pivmin) break;
  }
  if (tid == 0) lam_out[blockIdx.x] = 0.5 * (lo + hi);
}

// ------------------------------------------------------------------------------------------------
// K4b: eigenvectors of T for the selected eigenvalues by inverse iteration (LAPACK dstein's
// scheme: LU with partial pivoting of T - lambda I as in dgttrf, a few solves from a fixed
// pseudo-random start, re-orthogonalisation inside clusters).  The recurrences are serial in i,
// so lane 0 of a single wave runs them; norms and axpys use all 64 lanes.
// scratch: dl[n] dd[n] du[n] du2[n] y[n]; with use_lds the five arrays and the pivots live in
// dynamic LDS (44 n bytes, n <= ~3600), which cuts the latency of every step of the serial chain.
__global__ __launch_bounds__(64) void invit_kernel(const double* __restrict__ d, const double* __restrict__ e,
                                                   int n, const double* __restrict__ lam, int k,
                                                   double* __restrict__ z, double* __restrict__ scratch,
                                                   int* __restrict__ ipiv_global, int use_lds) {
  extern __shared__ __attribute__((aligned(16))) double invit_lds[];
  const int lane = threadIdx.x;
  double* base = use_lds ? invit_lds : scratch;
  double* dl = base;
  double* dd = base + (int64_t)n;
  double* du = base + 2 * (int64_t)n;
  double* du2 = base + 3 * (int64_t)n;
  double* y = base + 4 * (int64_t)n;
  int* ipiv = use_lds ? reinterpret_cast<int*>(invit_lds + 5 * (int64_t)n) : ipiv_global;

  double tn = 0.0;
  for (int i = lane; i < n; i += 64) {
    tn = fmax(tn, fabs(d[i]));
    if (i < n - 1) tn = fmax(tn, fabs(e[i]));
  }
  tn = wave_max(tn);
  tn = __shfl(tn, 0, 64);
  const double tiny = fmax(DBL_EPSILON * tn, DBL_MIN / DBL_EPSILON);
  const double ortol = 1e-3 * tn;

  // gridDim.x == 1: the k vectors one after the other (re-orthogonalisation inside clusters needs the earlier ones);
  // gridDim.x == k: one workgroup per vector -- the host launches that form when no two selected eigenvalues are
  // within ortol of each other, so that no vector waits for another
  const int c_begin = (gridDim.x == 1) ? 0 : (int)blockIdx.x;
  const int c_end = (gridDim.x == 1) ? k : (int)blockIdx.x + 1;
  for (int c = c_begin; c < c_end; ++c) {
    double* zc = z + (int64_t)c * n;
    const double lc = lam[c];
    if (n == 1) {
      if (lane == 0) zc[0] = 1.0;
      continue;
    }
    // start vector: fixed LCG stream in [-1, 1)
    for (int i = lane; i < n; i += 64) {
      uint32_t s = (uint32_t)(i + 1) * 2654435761u + (uint32_t)(c + 1) * 40503u;
      s = s * 1103515245u + 12345u;
      s ^= s >> 15;
      s = s * 1103515245u + 12345u;
      zc[i] = (double)(s >> 8) * (1.0 / 8388608.0) - 1.0;
    }
    __syncthreads();
    if (lane == 0) {
      // factor T - lc I = P L U  (dgttrf)
      for (int i = 0; i < n; ++i) {
        dd[i] = d[i] - lc;
        if (i < n - 1) { dl[i] = e[i]; du[i] = e[i]; }
        du2[i] = 0.0;
        ipiv[i] = i;
      }
      for (int i = 0; i < n - 1; ++i) {
        if (fabs(dd[i]) >= fabs(dl[i])) {
          if (dd[i] != 0.0) {
            const double f = dl[i] / dd[i];
            dl[i] = f;
            dd[i + 1] -= f * du[i];
          }
        } else {
          const double f = dd[i] / dl[i];
          dd[i] = dl[i];
          dl[i] = f;
          const double t = du[i];
          du[i] = dd[i + 1];
          dd[i + 1] = t - f * dd[i + 1];
          if (i < n - 2) {
            du2[i] = du[i + 1];
            du[i + 1] = -f * du[i + 1];
          }
          ipiv[i] = i + 1;
        }
      }
      for (int i = 0; i < n; ++i) {
        if (fabs(dd[i]) < tiny) dd[i] = (dd[i] < 0.0) ? -tiny : tiny;
      }
    }
    __syncthreads();
    for (int it = 0; it < 3; ++it) {
      if (lane == 0) {
        // y = U^-1 L^-1 P z  (dgtts2, no transpose)
        for (int i = 0; i < n; ++i) y[i] = zc[i];
        for (int i = 0; i < n - 1; ++i) {
          if (ipiv[i] == i) {
            y[i + 1] -= dl[i] * y[i];
          } else {
            const double t = y[i];
            y[i] = y[i + 1];
            y[i + 1] = t - dl[i] * y[i];
          }
        }
        y[n - 1] = y[n - 1] / dd[n - 1];
        if (n > 1) y[n - 2] = (y[n - 2] - du[n - 2] * y[n - 1]) / dd[n - 2];
        for (int i = n - 3; i >= 0; --i) y[i] = (y[i] - du[i] * y[i + 1] - du2[i] * y[i + 2]) / dd[i];
      }
      __syncthreads();
      // scale by max |y| first (the solve can grow by 1/eps), then orthogonalise + normalise
      double mx = 0.0;
      for (int i = lane; i < n; i += 64) mx = fmax(mx, fabs(y[i]));
      mx = wave_max(mx);
      mx = __shfl(mx, 0, 64);
      const double inv = (mx > 0.0) ? 1.0 / mx : 1.0;
      for (int i = lane; i < n; i += 64) y[i] *= inv;
      __syncthreads();
      for (int p = (gridDim.x == 1) ? 0 : c; p < c; ++p) {
        if (fabs(lam[p] - lc) <= ortol) {
          const double* zp = z + (int64_t)p * n;
          double dot = 0.0;
          for (int i = lane; i < n; i += 64) dot += zp[i] * y[i];
          dot = wave_sum(dot);
          dot = __shfl(dot, 0, 64);
          for (int i = lane; i < n; i += 64) y[i] -= dot * zp[i];
          __syncthreads();
        }
      }
      double nrm = 0.0;
      for (int i = lane; i < n; i += 64) nrm += y[i] * y[i];
      nrm = wave_sum(nrm);
      nrm = __shfl(nrm, 0, 64);
      const double rn = (nrm > 0.0) ? 1.0 / sqrt(nrm) : 0.0;
      for (int i = lane; i < n; i += 64) zc[i] = y[i] * rn;
      __syncthreads();
    }
  }
}

// ------------------------------------------------------------------------------------------------
// K5: eigenvector of B = Q z with Q = H_0 H_1 ... H_{n-3}: apply the reflectors in reverse order,
//   z <- z - tau_s (v_s . z) v_s ,  s = n-3 .. 0        (v_s in row s of a, columns s+1 .. n-1)
// One workgroup per eigenvector; thread t owns entries t, t+1024, ...  Then normalise and
// sign-normalise (largest-magnitude entry positive, ties -> lowest index).
__global__ __launch_bounds__(HW_T) void backtransform_kernel(const double* __restrict__ a, int n,
                                                             const double* __restrict__ tau,
                                                             double* __restrict__ z, int sign_normalize,
                                                             int apply_reflectors, double* __restrict__ out) {
  __shared__ double red[24];
  __shared__ int redi[24];
  const int tid = threadIdx.x;
  double* zc = z + (int64_t)blockIdx.x * n;
  for (int s = apply_reflectors ? n - 3 : -1; s >= 0; --s) {
    const double t = tau[s];
    if (t == 0.0) continue;  // uniform
    const double* vs = a + (int64_t)s * n;
    // Fixed ownership (entry j always belongs to thread j % 1024) so that a thread only ever
    // re-reads entries it wrote itself: no barrier is needed between consecutive reflectors.
    double part = 0.0;
    for (int j = tid; j < n; j += HW_T)
      if (j > s) part += vs[j] * zc[j];
    const double dot = block_reduce<0>(part, red);
    const double f = t * dot;
    for (int j = tid; j < n; j += HW_T)
      if (j > s) zc[j] -= f * vs[j];
  }
  double part = 0.0, amax = -1.0;
  int imax = 0x7fffffff;
  for (int j = tid; j < n; j += HW_T) {
    const double v = zc[j];
    part += v * v;
    if (fabs(v) > amax) { amax = fabs(v); imax = j; }
  }
  const double nrm2 = block_reduce<0>(part, red);
  const double gmax = block_reduce<1>(amax, red);
  // lowest index among the entries attaining the maximum magnitude
  int cand = (amax == gmax) ? imax : 0x7fffffff;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cand = min(cand, __shfl_down(cand, o, 64));
  __syncthreads();
  if ((tid & 63) == 0) redi[tid >> 6] = cand;
  __syncthreads();
  if (tid == 0) {
    int best = 0x7fffffff;
    for (int w = 0; w < HW_T / 64; ++w) best = min(best, redi[w]);
    redi[16] = best;
  }
  __syncthreads();
  const int ibest = redi[16];
  double sgn = 1.0;
  if (sign_normalize && ibest < n && zc[ibest] < 0.0) sgn = -1.0;
  const double rn = (nrm2 > 0.0) ? sgn / sqrt(nrm2) : 0.0;
  __syncthreads();  // everyone has read zc[ibest] before it is rescaled
  for (int j = tid; j < n; j += HW_T) {
    const double v = zc[j] * rn;
    zc[j] = v;
    out[(int64_t)blockIdx.x * n + j] = v;
  }
}

// ------------------------------------------------------------------------------------------------
// K5 blocked: Q z for the n-2 reflectors in blocks of BT = 32 (compact WY, LAPACK dlarft / dlarfb):
//   H_s0 H_s0+1 ... H_s0+31 = I - V T V^T,  V = the 32 reflector rows, T upper triangular,
//   T(i,i) = tau_i,  T(0:i, i) = -tau_i T(0:i, 0:i) (V(0:i,:) v_i).
// The serial kernel above applies 2,502 reflectors one after the other (2.1 us each: a dot product and a barrier per
// reflector, 5.3 ms at N = 2504).  Here the Gram matrices V V^T and the T factors of ALL blocks are built in two
// launches up front (they do not depend on z); then every block costs two short launches over column slices of 256:
// partial y = V z per slice, and (reduce y, t = T y, z -= V^T t) per slice.
constexpr int BT = 32;

// G[b][i][j] = v_{s0+i} . v_{s0+j} for j < i (one wave per (i, j) pair of a block; reflector s lives in row s of a,
// columns s+1 .. n-1, with an implicit 1 at s+1 that the tridiagonalisation stores explicitly)
__global__ __launch_bounds__(256) void wy_gram_kernel(const double* __restrict__ a, int n, int nref, double* __restrict__ g) {
  const int lane = threadIdx.x & 63;
  const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);   // 0 .. BT*BT-1
  const int blk = blockIdx.y;
  const int i = pair / BT, j = pair % BT;
  const int s0 = blk * BT;
  if (j >= i || s0 + i >= nref) return;
  const double* vi = a + (int64_t)(s0 + i) * n;
  const double* vj = a + (int64_t)(s0 + j) * n;
  double acc = 0.0;
  for (int c = s0 + i + 1 + lane; c < n; c += 64) acc += vi[c] * vj[c];   // v_i is zero up to column s0+i
  acc = wave_sum(acc);
  if (lane == 0) g[((int64_t)blk * BT + i) * BT + j] = acc;
}

// T factor of one block from its Gram matrix and tau (one wave per block)
__global__ __launch_bounds__(64) void wy_tfactor_kernel(const double* __restrict__ g, const double* __restrict__ tau, int nref,
                                                        double* __restrict__ t) {
  __shared__ double ts[BT][BT + 1];
  const int lane = threadIdx.x;
  const int blk = blockIdx.x;
  const int s0 = blk * BT;
  const int m = (nref - s0 < BT) ? (nref - s0) : BT;
  for (int e2 = lane; e2 < BT * BT; e2 += 64) ts[e2 / BT][e2 % BT] = 0.0;
  __syncthreads();
  for (int i = 0; i < m; ++i) {
    const double ti = tau[s0 + i];
    // column i: T(0:i, i) = -tau_i * T(0:i, 0:i) * G(i, 0:i)^T ; lane r computes row r
    double val = 0.0;
    if (lane < i) {
      for (int c = lane; c < i; ++c) val += ts[lane][c] * g[((int64_t)blk * BT + i) * BT + c];   // T upper triangular: c >= row
      val *= -ti;
    }
    __syncthreads();
    if (lane < i) ts[lane][i] = val;
    if (lane == i) ts[i][i] = ti;
    __syncthreads();
  }
  for (int e2 = lane; e2 < BT * BT; e2 += 64) t[(int64_t)blk * BT * BT + e2] = ts[e2 / BT][e2 % BT];
}

// partial y[slice][vec][i] = sum over the slice's columns of v_{s0+i}[c] * z_vec[c]
__global__ __launch_bounds__(256) void wy_dots_kernel(const double* __restrict__ a, int n, int nref, int blk,
                                                      const double* __restrict__ z, int k, double* __restrict__ ypart) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int slice = blockIdx.x, vec = blockIdx.y;
  const int s0 = blk * BT;
  const int c0 = slice * 256, c1 = (c0 + 256 < n) ? c0 + 256 : n;
  const double* zc = z + (int64_t)vec * n;
  for (int i = wave; i < BT; i += 4) {
    double acc = 0.0;
    if (s0 + i < nref) {
      const double* vi = a + (int64_t)(s0 + i) * n;
      for (int c = c0 + lane; c < c1; c += 64)
        if (c > s0 + i) acc += vi[c] * zc[c];
    }
    acc = wave_sum(acc);
    if (lane == 0) ypart[((int64_t)slice * k + vec) * BT + i] = acc;
  }
}

// z[c] -= sum_i v_{s0+i}[c] * (T y)[i] for the slice's columns; y is first reduced over the slices (fixed order)
__global__ __launch_bounds__(256) void wy_apply_kernel(const double* __restrict__ a, int n, int nref, int blk, int nslices,
                                                       const double* __restrict__ t, const double* __restrict__ ypart,
                                                       double* __restrict__ z, int k) {
  __shared__ double y[BT], ty[BT];
  const int slice = blockIdx.x, vec = blockIdx.y;
  const int s0 = blk * BT;
  if (threadIdx.x < BT) {
    double acc = 0.0;
    for (int sl = 0; sl < nslices; ++sl) acc += ypart[((int64_t)sl * k + vec) * BT + threadIdx.x];
    y[threadIdx.x] = acc;
  }
  __syncthreads();
  if (threadIdx.x < BT) {
    const double* tb = t + (int64_t)blk * BT * BT;
    double acc = 0.0;
    for (int c = threadIdx.x; c < BT; ++c) acc += tb[threadIdx.x * BT + c] * y[c];   // upper triangular
    ty[threadIdx.x] = acc;
  }
  __syncthreads();
  const int c = slice * 256 + threadIdx.x;
  if (c >= n) return;
  double* zc = z + (int64_t)vec * n;
  double acc = 0.0;
#pragma unroll 8
  for (int i = 0; i < BT; ++i)
    if (s0 + i < nref && c > s0 + i) acc += a[(int64_t)(s0 + i) * n + c] * ty[i];
  zc[c] -= acc;
}

}  // namespace

hipError_t launch_tridiagonalize(const EigWorkspace& ws, int32_t n, hipStream_t stream) {
  if (n <= 0) return hipErrorInvalidValue;
  if (n == 1) {
    hipLaunchKernelGGL(tridiag_hw_kernel, dim3(1), dim3(HW_T), 0, stream, ws.a, n, 0, ws.d, ws.e, ws.tau, ws.q,
                       ws.w);
    return hipGetLastError();
  }
  // Fused form (one launch per column) while v_{k-1}, w_{k-1}, v_k fit the LDS: n <= 6,800.  ws.w is 2 n doubles and
  // ws.scratch 6 n: q is double-buffered in ws.w, the current reflector in ws.scratch.
  const size_t lds_full = 3 * (size_t)n * sizeof(double);
  int kf = 0;  // columns [0, kf) take the fused kernel
  if (n >= 8 && lds_full <= 160 * 1024 - 1024) {
    if (lds_full > 64 * 1024) {
      static bool raised = false;  // opt in to more than 64 KiB of dynamic LDS once
      if (!raised) {
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(tridiag_fused_kernel),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);  // + 192 B static
        if (err != hipSuccess) return err;
        raised = true;
      }
    }
    kf = n - 2;  // steps 0 .. n-3 generate a reflector; step n-2 only closes d / e (tridiag_hw_kernel below)
    double* qb[2] = {ws.w, ws.w + n};
    double* vb[2] = {ws.scratch, ws.scratch + n};
    for (int k = 0; k < kf; ++k) {
      const int rows = n - k - 1;
      int nb = (rows + FT / 64 - 1) / (FT / 64);   // at least one row per wave ...
      if (nb > 256) nb = 256;                      // ... at most one workgroup per CU: every workgroup re-reads q and v
      nb = (nb + 7) / 8 * 8;                       // a multiple of the XCD count keeps row -> XCD fixed
      const size_t lds = 3 * (size_t)(n - k) * sizeof(double);
      hipLaunchKernelGGL(tridiag_fused_kernel, dim3((unsigned)nb), dim3(FT), lds, stream, ws.a, n, k, ws.d, ws.e, ws.tau,
                         qb[(k + 1) & 1], qb[k & 1], vb[(k + 1) & 1], vb[k & 1]);
    }
    // hand over to the closing step: it expects v_{n-3} in row n-3 and the raw mat-vec of step n-3 in ws.q
    hipError_t err = hipMemcpyAsync(ws.a + (int64_t)(kf - 1) * n + kf, vb[(kf - 1) & 1] + kf, sizeof(double) * (size_t)(n - kf),
                                    hipMemcpyDeviceToDevice, stream);
    if (err != hipSuccess) return err;
    err = hipMemcpyAsync(ws.q, qb[(kf - 1) & 1], sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, stream);
    if (err != hipSuccess) return err;
  }
  for (int k = kf; k <= n - 2; ++k) {
    const int hw_threads = HW_T;  // 256 threads were measured slower (7.6 vs 6.0 us per step at N = 2504)
    hipLaunchKernelGGL(tridiag_hw_kernel, dim3(1), dim3(hw_threads), 0, stream, ws.a, n, k, ws.d, ws.e, ws.tau,
                       ws.q, ws.w);
    if (k <= n - 3) {
      const int rows = n - k - 1;
      hipLaunchKernelGGL(tridiag_update_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, ws.a, n,
                         k, ws.w, ws.q);
    }
  }
  return hipGetLastError();
}

hipError_t launch_bisect(const EigWorkspace& ws, int32_t n, const int32_t* idx_host, int32_t count,
                         double* lam_out_dev, hipStream_t stream) {
  if (count <= 0) return hipSuccess;
  hipError_t err = hipMemcpyAsync(ws.iscratch, idx_host, sizeof(int32_t) * count, hipMemcpyHostToDevice, stream);
  if (err != hipSuccess) return err;
  const size_t lds_full = sizeof(double) * (32 + 2 * (size_t)n);
  const int use_lds = lds_full <= 64 * 1024;
  const size_t lds = use_lds ? lds_full : sizeof(double) * 32;
  hipLaunchKernelGGL(bisect_kernel, dim3((unsigned)count), dim3(256), lds, stream, ws.d, ws.e, n, ws.iscratch,
                     lam_out_dev, use_lds);
  return hipGetLastError();
}

namespace {
// dynamic LDS of invit_kernel for an n x n tridiagonal, or 0 when its five arrays go to global scratch
hipError_t invit_lds_bytes(int32_t n, size_t* lds_out) {
  const size_t lds = (size_t)n * (5 * sizeof(double) + sizeof(int)) + 16;
  const bool use_lds = lds <= 150 * 1024;
  if (use_lds && lds > 64 * 1024) {
    static bool raised = false;  // opt in to more than 64 KiB of dynamic LDS once
    if (!raised) {
      hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(invit_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      if (err != hipSuccess) return err;
      raised = true;
    }
  }
  *lds_out = use_lds ? lds : 0;
  return hipSuccess;
}
}  // namespace

hipError_t launch_inverse_iteration(const EigWorkspace& ws, int32_t n, const double* lam_sel_host, int32_t k,
                                    hipStream_t stream) {
  // selected eigenvalues go to ws.lam[0..k) (the candidates there have been consumed by the host)
  hipError_t err = hipMemcpyAsync(ws.lam, lam_sel_host, sizeof(double) * k, hipMemcpyHostToDevice, stream);
  if (err != hipSuccess) return err;
  size_t lds = 0;
  if ((err = invit_lds_bytes(n, &lds)) != hipSuccess) return err;
  const int use_lds = lds > 0;
  // One workgroup per vector when each has its own scratch (LDS) and no two selected eigenvalues can fall into one
  // cluster (the kernel's ortol = 1e-3 max|T|; |T| <= 3 max(|d|, |e|) is not known here, so the test is against the
  // largest selected |lambda| -- a lower bound of max|T| would be the unsafe direction, this one is an upper bound of
  // nothing: hence the conservative factor 1e-2)
  bool separate = use_lds && k > 1;
  if (separate) {
    double big = 0.0;
    for (int c = 0; c < k; ++c) big = std::max(big, std::fabs(lam_sel_host[c]));
    for (int c = 0; c < k && separate; ++c)
      for (int p = 0; p < c; ++p)
        if (std::fabs(lam_sel_host[c] - lam_sel_host[p]) <= 1e-2 * big) { separate = false; break; }
  }
  hipLaunchKernelGGL(invit_kernel, dim3(separate ? (unsigned)k : 1u), dim3(64), lds, stream, ws.d, ws.e, n,
                     ws.lam, k, ws.z, ws.scratch, ws.iscratch, use_lds);
  return hipGetLastError();
}

hipError_t launch_inverse_iteration_dev(const EigWorkspace& ws, int32_t n, int32_t k, hipStream_t stream) {
  size_t lds = 0;
  hipError_t err = invit_lds_bytes(n, &lds);
  if (err != hipSuccess) return err;
  hipLaunchKernelGGL(invit_kernel, dim3(1), dim3(64), lds, stream, ws.d, ws.e, n, ws.lam, k, ws.z, ws.scratch,
                     ws.iscratch, lds > 0 ? 1 : 0);
  return hipGetLastError();
}

hipError_t launch_backtransform(const EigWorkspace& ws, int32_t n, int32_t k, int sign_normalize,
                                int apply_reflectors, double* out_dev, hipStream_t stream) {
  // blocked form when a WY workspace exists (ws.wy: nblk * (2 * BT * BT) + nslices * k * BT doubles) and there are enough
  // reflectors for it to pay; the serial kernel then only normalises
  const int nref = n - 2;
  if (apply_reflectors && ws.wy != nullptr && nref >= 4 * BT) {
    const int nblk = (nref + BT - 1) / BT;
    const int nslices = (n + 255) / 256;
    double* g = ws.wy;
    double* t = g + (int64_t)nblk * BT * BT;
    double* ypart = t + (int64_t)nblk * BT * BT;
    hipLaunchKernelGGL(wy_gram_kernel, dim3(BT * BT / 4, (unsigned)nblk), dim3(256), 0, stream, ws.a, n, nref, g);
    hipLaunchKernelGGL(wy_tfactor_kernel, dim3((unsigned)nblk), dim3(64), 0, stream, g, ws.tau, nref, t);
    for (int blk = nblk - 1; blk >= 0; --blk) {   // Q z = H_0 ( H_1 ( ... H_{n-3} z)): the last block acts first
      hipLaunchKernelGGL(wy_dots_kernel, dim3((unsigned)nslices, (unsigned)k), dim3(256), 0, stream, ws.a, n, nref, blk, ws.z,
                         k, ypart);
      hipLaunchKernelGGL(wy_apply_kernel, dim3((unsigned)nslices, (unsigned)k), dim3(256), 0, stream, ws.a, n, nref, blk,
                         nslices, t, ypart, ws.z, k);
    }
    apply_reflectors = 0;
  }
  hipLaunchKernelGGL(backtransform_kernel, dim3((unsigned)k), dim3(HW_T), 0, stream, ws.a, n, ws.tau, ws.z,
                     sign_normalize, apply_reflectors, out_dev);
  return hipGetLastError();
}

size_t wy_workspace_doubles(int32_t n, int32_t k) {
  const int64_t nblk = ((int64_t)n - 2 + BT - 1) / BT;
  const int64_t nslices = ((int64_t)n + 255) / 256;
  return (size_t)(std::max<int64_t>(nblk, 1) * 2 * BT * BT + nslices * k * BT);
}

}  // namespace pcoa
