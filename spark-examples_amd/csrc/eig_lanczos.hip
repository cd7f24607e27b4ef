// eig_lanczos.hip -- top-k eigenpairs of the centred matrix B by Lanczos with full
// re-orthogonalisation: the fast path of computePca part 3 (reference VariantsPca.scala:224-227,
// MLlib RowMatrix.computePrincipalComponents).  Only k = numPc (2) eigenpairs are wanted, so instead
// of reducing all of B to tridiagonal form (eig.hip, O(N^3), ~2500 serial steps) a Krylov basis of a
// few dozen vectors is built; every step is one read of B (50 MB fp64 at N = 2504, resident in the
// 256 MB Infinity Cache):
//
//   w = B v_j                                   symv_kernel        one wave per row, HBM/MALL-bound
//   w -= V_j (V_j^T w)   twice (CGS2)           cgs_dots / cgs_update
//   alpha_j = v_j^T B v_j, beta_j = ||w||, v_{j+1} = w / beta_j      lanczos_finish_kernel
//
// The small tridiagonal T_m (alpha, beta) goes through the SAME bisection + inverse-iteration
// kernels as the dense path (eig.hip) to give Ritz values theta and vectors y; the residual of a
// Ritz pair is |beta_m y_m| and costs nothing.  When every wanted pair has converged the Ritz
// vectors u = V_m y are formed and the TRUE residual ||B u - theta u|| is measured with one more
// symv; only a pair that passes that test is returned.  Anything else (slow convergence because of
// a tiny spectral gap, breakdown) makes the caller fall back to the Householder solver, so the fast
// path can never return a wrong answer silently.
#include <algorithm>
#include <cfloat>
#include <cstdio>
#include <cstdlib>

#include "pcoa_internal.h"

namespace pcoa {
namespace {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// block-wide sum broadcast to all threads; red >= 17 doubles
__device__ __forceinline__ double block_sum(double v, double* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  if (threadIdx.x < 64) {
    double t = ((int)threadIdx.x < nw) ? red[threadIdx.x] : 0.0;
    t = wave_sum(t);
    if (threadIdx.x == 0) red[16] = t;
  }
  __syncthreads();
  return red[16];
}

// v0 = normalised fixed pseudo-random vector (single workgroup)
__global__ __launch_bounds__(1024) void lanczos_init_kernel(double* __restrict__ v0, int n) {
  __shared__ double red[24];
  double part = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) {
    uint32_t s = (uint32_t)(i + 1) * 2654435761u + 12345u;
    s = s * 1103515245u + 12345u;
    s ^= s >> 15;
    s = s * 1103515245u + 12345u;
    const double x = (double)(s >> 8) * (1.0 / 8388608.0) - 1.0;
    v0[i] = x;
    part += x * x;
  }
  const double nrm2 = block_sum(part, red);
  const double rn = 1.0 / sqrt(nrm2);
  for (int i = threadIdx.x; i < n; i += 1024) v0[i] *= rn;
}

// y = A x, A symmetric dense row-major: one wave per row, 8 loads in flight per lane
__global__ __launch_bounds__(256) void symv_kernel(const double* __restrict__ a, int n,
                                                   const double* __restrict__ x, double* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const double* row = a + (int64_t)i * n;
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
  int j = lane;
  for (; j + 448 < n; j += 512) {
    const double r0 = row[j], r1 = row[j + 64], r2 = row[j + 128], r3 = row[j + 192];
    const double r4 = row[j + 256], r5 = row[j + 320], r6 = row[j + 384], r7 = row[j + 448];
    acc0 += r0 * x[j] + r4 * x[j + 256];
    acc1 += r1 * x[j + 64] + r5 * x[j + 320];
    acc2 += r2 * x[j + 128] + r6 * x[j + 384];
    acc3 += r3 * x[j + 192] + r7 * x[j + 448];
  }
  for (; j < n; j += 64) acc0 += row[j] * x[j];
  const double acc = wave_sum((acc0 + acc1) + (acc2 + acc3));
  if (lane == 0) y[i] = acc;
}

// y = B x with B evaluated on the fly from the integer similarity matrix (EigWorkspace, implicit form): the same
// loop structure as symv_kernel, the same per-entry expression as center_kernel (no fused multiply-add in the
// centring), so the result equals symv_kernel on the materialised B bit for bit -- at 4 (12) instead of 8 bytes per
// entry and without the N x N fp64 matrix (80 GB at N = 100,000).
template <bool HAS64>
__global__ __launch_bounds__(256) void symv_centered_kernel(const int32_t* __restrict__ s32,
                                                            const int64_t* __restrict__ s64, int n,
                                                            const double* __restrict__ cm,
                                                            const double* __restrict__ stats,
                                                            const double* __restrict__ x, double* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int64_t base = (int64_t)i * n;
  const double row_mean = cm[i];
  const double mmean = stats[1];
  auto entry = [&](int j) -> double {
#pragma clang fp contract(off)
    const double data = HAS64 ? (double)((int64_t)s32[base + j] + s64[base + j]) : (double)s32[base + j];
    double t = data - row_mean;
    t = t - cm[j];
    t = t + mmean;
    return t;
  };
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
  int j = lane;
  for (; j + 448 < n; j += 512) {
    const double r0 = entry(j), r1 = entry(j + 64), r2 = entry(j + 128), r3 = entry(j + 192);
    const double r4 = entry(j + 256), r5 = entry(j + 320), r6 = entry(j + 384), r7 = entry(j + 448);
    acc0 += r0 * x[j] + r4 * x[j + 256];
    acc1 += r1 * x[j + 64] + r5 * x[j + 320];
    acc2 += r2 * x[j + 128] + r6 * x[j + 384];
    acc3 += r3 * x[j + 192] + r7 * x[j + 448];
  }
  for (; j < n; j += 64) acc0 += entry(j) * x[j];
  const double acc = wave_sum((acc0 + acc1) + (acc2 + acc3));
  if (lane == 0) y[i] = acc;
}

void launch_symv(const EigWorkspace& ws, int n, const double* x, double* y, hipStream_t stream) {
  const unsigned rows4 = (unsigned)((n + 3) / 4);
  if (ws.a) {
    hipLaunchKernelGGL(symv_kernel, dim3(rows4), dim3(256), 0, stream, ws.a, n, x, y);
  } else if (ws.s64) {
    hipLaunchKernelGGL(symv_centered_kernel<true>, dim3(rows4), dim3(256), 0, stream, ws.s32, ws.s64, n, ws.colmean,
                       ws.stats, x, y);
  } else {
    hipLaunchKernelGGL(symv_centered_kernel<false>, dim3(rows4), dim3(256), 0, stream, ws.s32, ws.s64, n, ws.colmean,
                       ws.stats, x, y);
  }
}

// h[p] = V[p] . w   for p = 0 .. count-1 (one workgroup per p)
__global__ __launch_bounds__(256) void cgs_dots_kernel(const double* __restrict__ v, int n,
                                                       const double* __restrict__ w, double* __restrict__ h) {
  __shared__ double red[24];
  const double* vp = v + (int64_t)blockIdx.x * n;
  double part = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) part += vp[i] * w[i];
  const double t = block_sum(part, red);
  if (threadIdx.x == 0) h[blockIdx.x] = t;
}

// w[i] -= sum_p h[p] V[p][i]
__global__ __launch_bounds__(256) void cgs_update_kernel(const double* __restrict__ v, int n, int count,
                                                         const double* __restrict__ h, double* __restrict__ w) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double acc = 0.0;
  for (int p = 0; p < count; ++p) acc += h[p] * v[(int64_t)p * n + i];
  w[i] -= acc;
}

// alpha[j] = h1[j] + h2[j]; beta[j] = ||w||; V[j+1] = w / beta[j]   (single workgroup)
__global__ __launch_bounds__(1024) void lanczos_finish_kernel(double* __restrict__ v, int n, int j,
                                                              const double* __restrict__ w,
                                                              const double* __restrict__ h1,
                                                              const double* __restrict__ h2,
                                                              double* __restrict__ alpha, double* __restrict__ beta) {
  __shared__ double red[24];
  double part = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) part += w[i] * w[i];
  const double nrm = sqrt(block_sum(part, red));
  const double rn = (nrm > 0.0) ? 1.0 / nrm : 0.0;
  double* vn = v + (int64_t)(j + 1) * n;
  for (int i = threadIdx.x; i < n; i += 1024) vn[i] = w[i] * rn;
  if (threadIdx.x == 0) {
    alpha[j] = h1[j] + h2[j];
    beta[j] = nrm;
  }
}

// u[c][i] = sum_p y[c][p] V[p][i]    (Ritz vectors), grid (ceil(n/256), k)
__global__ __launch_bounds__(256) void ritz_kernel(const double* __restrict__ v, int n, int m,
                                                   const double* __restrict__ y, double* __restrict__ u) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int c = blockIdx.y;
  if (i >= n) return;
  const double* yc = y + (int64_t)c * m;
  double acc = 0.0;
  for (int p = 0; p < m; ++p) acc += yc[p] * v[(int64_t)p * n + i];
  u[(int64_t)c * n + i] = acc;
}

// res[c] = ||bu - theta[c] u[c]|| / ||u[c]||   with bu = B u[c] (one workgroup per c)
__global__ __launch_bounds__(1024) void residual_kernel(const double* __restrict__ bu, const double* __restrict__ u,
                                                        int n, const double* __restrict__ theta,
                                                        double* __restrict__ res) {
  __shared__ double red[24];
  const int c = blockIdx.x;
  const double th = theta[c];
  double pr = 0.0, pu = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) {
    const double ui = u[(int64_t)c * n + i];
    const double r = bu[(int64_t)c * n + i] - th * ui;
    pr += r * r;
    pu += ui * ui;
  }
  const double r2 = block_sum(pr, red);
  const double u2 = block_sum(pu, red);
  if (threadIdx.x == 0) res[c] = (u2 > 0.0) ? sqrt(r2 / u2) : DBL_MAX;
}

}  // namespace

size_t lanczos_workspace_doubles(int32_t n, int32_t k, int32_t mmax) {
  // V[(mmax+1)][n], w[n], bu[k][n], alpha[mmax], beta[mmax], h1[mmax+1], h2[mmax+1], small: lam[2k+2], res[k]
  return (size_t)(mmax + 1) * n + (size_t)n + (size_t)k * n + 4 * (size_t)(mmax + 2) + 2 * (size_t)k + 8;
}

// Returns hipSuccess on a clean run; *converged tells whether ws.z[0..k) holds verified eigenvectors of
// B (unnormalised Ritz vectors; the caller normalises) and lam_sel_host[0..k) their eigenvalues
// (ordered by decreasing magnitude).  B (ws.a, or the implicit form when ws.a is null) is not modified.
hipError_t lanczos_topk(const EigWorkspace& ws, double* lz, int32_t n, int32_t k, int32_t mmax, double tol,
                        double* lam_sel_host, int* converged, int* steps_out, hipStream_t stream) {
  *converged = 0;
  if (steps_out) *steps_out = 0;
  if (mmax > n) mmax = n;
  if (mmax < k + 2 || n < 8) return hipSuccess;  // tiny problems go to the dense path
  double* V = lz;
  double* w = V + (size_t)(mmax + 1) * n;
  double* bu = w + n;
  double* alpha = bu + (size_t)k * n;
  double* beta = alpha + (mmax + 2);
  double* h1 = beta + (mmax + 2);
  double* h2 = h1 + (mmax + 2);
  double* res = h2 + (mmax + 2);

  EigWorkspace small = ws;  // T_m goes through the dense path's bisection / inverse iteration
  small.d = alpha;
  small.e = beta;

  hipLaunchKernelGGL(lanczos_init_kernel, dim3(1), dim3(1024), 0, stream, V, n);
  const unsigned nb = (unsigned)((n + 255) / 256);
  // Ritz pairs are examined at m = 12, 16, 20, 24, then every 8 steps up to 64, then every m / 2: a check costs about
  // four steps (bisection + inverse iteration on T_m + two host round trips), and population structure converges
  // early (configs[1] stand-in: estimate 1e-14 at m = 12, true relative residual 3.7e-9)
  int next_check = 12;
  if (debug_knobs().lanczos_first_check > 0) next_check = std::max(4, debug_knobs().lanczos_first_check);
  if (next_check > mmax) next_check = mmax;
  std::vector<double> cand, ylast((size_t)k), hres((size_t)k);
  std::vector<int32_t> idx;
  for (int j = 0; j < mmax; ++j) {
    const double* vj = V + (size_t)j * n;
    launch_symv(ws, n, vj, w, stream);
    // (the five small kernels below fused into ONE workgroup were measured slower: 30 vs 24 us per step -- a single
    // CU cannot stream the ~2 MB of basis vectors a step touches as fast as j+1 workgroups can)
    hipLaunchKernelGGL(cgs_dots_kernel, dim3((unsigned)(j + 1)), dim3(256), 0, stream, V, n, w, h1);
    hipLaunchKernelGGL(cgs_update_kernel, dim3(nb), dim3(256), 0, stream, V, n, j + 1, h1, w);
    hipLaunchKernelGGL(cgs_dots_kernel, dim3((unsigned)(j + 1)), dim3(256), 0, stream, V, n, w, h2);
    hipLaunchKernelGGL(cgs_update_kernel, dim3(nb), dim3(256), 0, stream, V, n, j + 1, h2, w);
    hipLaunchKernelGGL(lanczos_finish_kernel, dim3(1), dim3(1024), 0, stream, V, n, j, w, h1, h2, alpha, beta);
    const int m = j + 1;
    if (m != next_check && m != mmax) continue;
    next_check = (m < 24) ? m + 4 : (m < 64) ? m + 8 : m + m / 2;
    if (next_check > mmax) next_check = mmax;
    if (steps_out) *steps_out = m;

    // Ritz values of T_m: k largest and k smallest, keep the k of largest magnitude (MLlib ranks by |lambda|)
    // (one extra value at each end so that the spectral gap of every kept pair can be estimated)
    idx.clear();
    for (int t = 0; t <= k && t < m; ++t) idx.push_back(m - 1 - t);
    for (int t = 0; t <= k; ++t)
      if (t < m - 1 - k) idx.push_back(t);
    cand.resize(idx.size());
    hipError_t e = launch_bisect(small, m, idx.data(), (int32_t)idx.size(), ws.lam, stream);
    if (e != hipSuccess) return e;
    double beta_m = 0.0;
    if ((e = hipMemcpyAsync(cand.data(), ws.lam, sizeof(double) * cand.size(), hipMemcpyDeviceToHost, stream)) !=
        hipSuccess)
      return e;
    if ((e = hipMemcpyAsync(&beta_m, beta + (m - 1), sizeof(double), hipMemcpyDeviceToHost, stream)) != hipSuccess)
      return e;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
    std::vector<int> order(cand.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
      const double fa = fabs(cand[a]), fb = fabs(cand[b]);
      if (fa != fb) return fa > fb;
      return cand[a] > cand[b];
    });
    double scale = 0.0;
    for (double c : cand) scale = fmax(scale, fabs(c));
    for (int t = 0; t < k; ++t) lam_sel_host[t] = cand[order[t]];
    if (!(scale > 0.0) || !std::isfinite(scale)) return hipSuccess;  // B == 0 or garbage: dense path decides
    // Ritz vectors y of T_m and the free residual estimate |beta_m * y_last|
    if ((e = launch_inverse_iteration(small, m, lam_sel_host, k, stream)) != hipSuccess) return e;
    for (int t = 0; t < k; ++t)
      if ((e = hipMemcpyAsync(&ylast[t], ws.z + (size_t)t * m + (m - 1), sizeof(double), hipMemcpyDeviceToHost,
                              stream)) != hipSuccess)
        return e;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
    // A Ritz pair is accepted when its residual is small against the spectrum AND against its own
    // gap (eigenvector error ~ residual / gap): target 1e-8, two orders inside the 1e-6 parity bar.
    std::vector<double> gap((size_t)k);
    for (int t = 0; t < k; ++t) {
      double g = DBL_MAX;
      for (size_t c2 = 0; c2 < cand.size(); ++c2)
        if ((int)c2 != order[t]) g = fmin(g, fabs(cand[c2] - lam_sel_host[t]));
      gap[(size_t)t] = g;
    }
    auto accept = [&](double r, int t) { return r <= tol * scale && r <= 1e-8 * gap[(size_t)t]; };
    bool ok = true;
    for (int t = 0; t < k; ++t) ok = ok && accept(fabs(beta_m * ylast[t]), t);
    const bool breakdown = beta_m <= 1e-14 * scale;  // invariant subspace: T_m holds exact eigenvalues
    const bool trace = debug_knobs().lanczos_trace != 0;
    if (trace) {
      std::fprintf(stderr, "[lanczos] m=%d beta_m=%.3e scale=%.6e", m, beta_m, scale);
      for (int t = 0; t < k; ++t)
        std::fprintf(stderr, "  theta%d=%.10e est=%.3e gap=%.3e", t, lam_sel_host[t], fabs(beta_m * ylast[t]), gap[(size_t)t]);
      std::fprintf(stderr, "  ok=%d\n", (int)ok);
    }
    if (!ok && !breakdown) {
      if (m == mmax) return hipSuccess;
      continue;
    }
    // u = V_m y, true residual with one more pass over B per vector
    hipLaunchKernelGGL(ritz_kernel, dim3(nb, (unsigned)k), dim3(256), 0, stream, V, n, m, ws.z, bu);
    // ws.z <- u (ritz_kernel read y from ws.z, so it wrote to bu first)
    if ((e = hipMemcpyAsync(ws.z, bu, sizeof(double) * (size_t)k * n, hipMemcpyDeviceToDevice, stream)) != hipSuccess)
      return e;
    for (int t = 0; t < k; ++t)
      launch_symv(ws, n, ws.z + (size_t)t * n, bu + (size_t)t * n, stream);
    hipLaunchKernelGGL(residual_kernel, dim3((unsigned)k), dim3(1024), 0, stream, bu, ws.z, n, ws.lam, res);
    if ((e = hipMemcpyAsync(hres.data(), res, sizeof(double) * k, hipMemcpyDeviceToHost, stream)) != hipSuccess)
      return e;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
    bool verified = true;
    for (int t = 0; t < k; ++t) verified = verified && std::isfinite(hres[t]) && accept(hres[t] * 0.125, t);
    if (trace) {
      std::fprintf(stderr, "[lanczos] m=%d true residuals:", m);
      for (int t = 0; t < k; ++t) std::fprintf(stderr, " %.3e", hres[t]);
      std::fprintf(stderr, "  verified=%d\n", (int)verified);
    }
    if (verified) {
      *converged = 1;
      return hipGetLastError();
    }
    if (breakdown || m == mmax) return hipSuccess;  // not trustworthy: dense path
  }
  return hipGetLastError();
}

}  // namespace pcoa
