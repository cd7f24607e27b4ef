// eig_lanczos.hip -- top-k eigenpairs of the centred matrix B by Lanczos with full
// re-orthogonalisation: the fast path of computePca part 3 (reference VariantsPca.scala:224-227,
// MLlib RowMatrix.computePrincipalComponents).  Only k = numPc (2) eigenpairs are wanted, so instead
// of reducing all of B to tridiagonal form (eig.hip, O(N^3), ~2500 serial steps) a Krylov basis of a
// few dozen vectors is built; every step is one read of B (50 MB fp64 at N = 2504, resident in the
// 256 MB Infinity Cache):
//
//   w = B v_j                                   symv_kernel / symv_centered_kernel   one wave per row, MALL-bound
//   h1 = V_j^T w                                cgs_dots_kernel          (pass 1 of CGS2)
//   w -= V_j h1;  slice shares of V_j^T w       cgs_update_dots_kernel   (pass 2's dots on the way out)
//   w -= V_j h2;  slice shares of ||w||^2       cgs_update_norm_kernel
//   alpha_j = h1_j + h2_j, beta_j = ||w||, v_{j+1} = w / beta_j      lanczos_finish_kernel
//
// The small tridiagonal T_m (alpha, beta) goes through the SAME bisection + inverse-iteration
// kernels as the dense path (eig.hip) to give Ritz values theta and vectors y; the residual of a
// Ritz pair is |beta_m y_m| and costs nothing.  The Ritz vectors u = V_m y and the TRUE residual
// ||B u - theta u|| (one more symv per vector) are queued behind it speculatively and the whole check is
// read back in one record; only a pair whose estimate AND true residual pass is returned.  Anything else (slow convergence because of
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
__global__ __launch_bounds__(1024) void lanczos_init_kernel(double* __restrict__ v0, int n, uint32_t salt = 0u) {
  __shared__ double red[24];
  double part = 0.0;
  for (int i = threadIdx.x; i < n; i += 1024) {
    uint32_t s = (uint32_t)(i + 1) * 2654435761u + 12345u + salt * 0x9E3779B9u;
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

// Row i of a symmetric mat-vec, one wave per row: sum_j entry(i, j) x[j], valid in lane 0.  `quad(j, out)` yields the
// four matrix entries of columns j .. j+3 (n % 4 == 0: 16- / 32-byte loads, four groups = 1024 columns of the row in
// flight per wave -- with 4-byte loads the wave had 2 KB in flight and the kernel ran at 1.9 TB/s out of the Infinity
// Cache); `one(j)` a single entry (any n).  The explicit and the implicit (centred on the fly) kernel share this
// function, i.e. the same association of every sum: their results are identical bit for bit when their entries are.
template <class Quad, class One>
__device__ __forceinline__ double row_dot(Quad quad, One one, const double* __restrict__ x, int n, int lane) {
  double acc0 = 0.0, acc1 = 0.0, acc2 = 0.0, acc3 = 0.0;
  if ((n & 3) == 0) {
    int j = 4 * lane;
    for (; j + 768 < n; j += 1024) {
      double e0[4], e1[4], e2[4], e3[4];
      quad(j, e0);
      quad(j + 256, e1);
      quad(j + 512, e2);
      quad(j + 768, e3);
      const double2 xa0 = *reinterpret_cast<const double2*>(x + j), xb0 = *reinterpret_cast<const double2*>(x + j + 2);
      const double2 xa1 = *reinterpret_cast<const double2*>(x + j + 256), xb1 = *reinterpret_cast<const double2*>(x + j + 258);
      const double2 xa2 = *reinterpret_cast<const double2*>(x + j + 512), xb2 = *reinterpret_cast<const double2*>(x + j + 514);
      const double2 xa3 = *reinterpret_cast<const double2*>(x + j + 768), xb3 = *reinterpret_cast<const double2*>(x + j + 770);
      acc0 += e0[0] * xa0.x + e2[0] * xa2.x;
      acc1 += e0[1] * xa0.y + e2[1] * xa2.y;
      acc2 += e0[2] * xb0.x + e2[2] * xb2.x;
      acc3 += e0[3] * xb0.y + e2[3] * xb2.y;
      acc0 += e1[0] * xa1.x + e3[0] * xa3.x;
      acc1 += e1[1] * xa1.y + e3[1] * xa3.y;
      acc2 += e1[2] * xb1.x + e3[2] * xb3.x;
      acc3 += e1[3] * xb1.y + e3[3] * xb3.y;
    }
    for (; j < n; j += 256) {
      double e0[4];
      quad(j, e0);
      const double2 xa0 = *reinterpret_cast<const double2*>(x + j), xb0 = *reinterpret_cast<const double2*>(x + j + 2);
      acc0 += e0[0] * xa0.x;
      acc1 += e0[1] * xa0.y;
      acc2 += e0[2] * xb0.x;
      acc3 += e0[3] * xb0.y;
    }
  } else {
    int j = lane;
    for (; j + 448 < n; j += 512) {
      const double r0 = one(j), r1 = one(j + 64), r2 = one(j + 128), r3 = one(j + 192);
      const double r4 = one(j + 256), r5 = one(j + 320), r6 = one(j + 384), r7 = one(j + 448);
      acc0 += r0 * x[j] + r4 * x[j + 256];
      acc1 += r1 * x[j + 64] + r5 * x[j + 320];
      acc2 += r2 * x[j + 128] + r6 * x[j + 384];
      acc3 += r3 * x[j + 192] + r7 * x[j + 448];
    }
    for (; j < n; j += 64) acc0 += one(j) * x[j];
  }
  return wave_sum((acc0 + acc1) + (acc2 + acc3));
}

// y = A x, A symmetric dense row-major: one wave per row
__global__ __launch_bounds__(256) void symv_kernel(const double* __restrict__ a, int n,
                                                   const double* __restrict__ x, double* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const double* row = a + (int64_t)i * n;
  auto quad = [&](int j, double* out) {
    const double2 lo = *reinterpret_cast<const double2*>(row + j), hi = *reinterpret_cast<const double2*>(row + j + 2);
    out[0] = lo.x; out[1] = lo.y; out[2] = hi.x; out[3] = hi.y;
  };
  auto one = [&](int j) -> double { return row[j]; };
  const double acc = row_dot(quad, one, x, n, lane);
  if (lane == 0) y[i] = acc;
}

// y = B x with B evaluated on the fly from the integer similarity matrix (EigWorkspace, implicit form): the same
// row_dot as symv_kernel, the same per-entry expression as center_kernel (no fused multiply-add in the
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
  auto centre = [&](double data, double col_mean) -> double {
#pragma clang fp contract(off)
    double t = data - row_mean;
    t = t - col_mean;
    t = t + mmean;
    return t;
  };
  auto quad = [&](int j, double* out) {
    const int4 s = *reinterpret_cast<const int4*>(s32 + base + j);
    double d0 = (double)s.x, d1 = (double)s.y, d2 = (double)s.z, d3 = (double)s.w;
    if (HAS64) {
      const longlong2 l = *reinterpret_cast<const longlong2*>(s64 + base + j);
      const longlong2 h = *reinterpret_cast<const longlong2*>(s64 + base + j + 2);
      d0 = (double)((int64_t)s.x + l.x); d1 = (double)((int64_t)s.y + l.y);
      d2 = (double)((int64_t)s.z + h.x); d3 = (double)((int64_t)s.w + h.y);
    }
    const double2 ca = *reinterpret_cast<const double2*>(cm + j), cb = *reinterpret_cast<const double2*>(cm + j + 2);
    out[0] = centre(d0, ca.x); out[1] = centre(d1, ca.y); out[2] = centre(d2, cb.x); out[3] = centre(d3, cb.y);
  };
  auto one = [&](int j) -> double {
    const double data = HAS64 ? (double)((int64_t)s32[base + j] + s64[base + j]) : (double)s32[base + j];
    return centre(data, cm[j]);
  };
  const double acc = row_dot(quad, one, x, n, lane);
  if (lane == 0) y[i] = acc;
}

// ---- large N: y = B x reading only the UPPER TRIANGLE of S (r04) ----------------------------------------
// symv_centered_kernel reads all N^2 entries per mat-vec -- 40 GB at N = 100,000, 250 GB at N = 250,000 -- and with one wave
// per row streams them at ~3 TB/s.  S is symmetric: an entry S(i, j), j > i, serves y_i += B(i,j) x_j AND y_j += B(j,i) x_i.
// One workgroup per upper-triangular tile of 1024 x 1024 entries (4 MiB of S): wave w walks the rows w, w + 4, ..; a lane owns
// 16 columns (four 16-byte loads per row, the next row's loads in flight while this row is used), accumulates its share of
// the row's dot product (wave-reduced at the end of the row) and, per column, the transposed products.  Both B(i,j) and
// B(j,i) are evaluated from the integer S in the reference's operation order -- ((S - rowMean) - colMean) + mean, no fused
// multiply-add in the centring (VariantsPca.scala:216-221) -- so the entries are the ones symv_centered_kernel and the
// materialised B hold, bit for bit; only the ORDER of the additions differs (results agree to ~1e-15 relative, tested to
// 1e-13).  No floating-point atomics: a tile writes its 1024 row sums and 1024 column sums into its own slots of `part`,
// and symv_sym_gather_kernel adds, for every y_i, the slots of its block row and block column in a fixed order.
constexpr int SYT = 1024;

__device__ __forceinline__ int64_t sym_tile_index(int bi, int bj, int nb) {  // bi <= bj
  return (int64_t)bi * nb - (int64_t)bi * (bi - 1) / 2 + (bj - bi);
}

// Sum over the 64 lanes on the VALU (DPP: xor 1, xor 2, mirror within 8, mirror within 16, then row 0 -> 1 / 2 -> 3 and
// rows 0-1 -> 2-3 broadcasts); the total is in LANE 63.  One dependent chain of 18 VALU instructions instead of six LDS
// round trips (ds_bpermute) per row.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int l2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);
  const int h2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
  return v + __hiloint2double(h2, l2);
}
__device__ __forceinline__ double wave_sum_to_lane63(double v) {
  v = dpp_add<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141, 0xF>(v);  // row_half_mirror
  v = dpp_add<0x140, 0xF>(v);  // row_mirror: every lane holds the sum of its row of 16
  v = dpp_add<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
  v = dpp_add<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3
  return v;
}

// r05: what the r04 form lost (4.4 TB/s): the row's x_i / rowMean_i were vector loads issued BEHIND the next row's
// prefetch, and vector-memory returns are counted in order -- waiting for them waited for the prefetch as well
// (s_waitcnt vmcnt(0) in every iteration); the `j < n` masks of the last block column were evaluated for every tile
// (16 exec-mask branches per row); the row sum went through six LDS round trips; a lane's 16 columns cost 96 registers
// of loop invariants (152 in all: three waves per SIMD).  Here the four waves of a workgroup are a 2 x 2 grid -- wave
// (wr, wc) walks the rows of parity wr over the 512 columns of half wc, a lane owns 8 columns (48 registers of
// invariants) --, the tile's 1024 x_i and rowMean_i are staged in LDS once, rows are prefetched FOUR deep (8 x 1 KiB per
// wave in flight; the loop holds no vector-memory instruction but the 16-byte loads of S), interior tiles carry no masks
// (EDGE is a per-workgroup branch), row sums are reduced with DPP and collected in LDS.  Nothing is combined across
// waves inside the kernel: a tile writes the row sums of its two column halves and the column sums of its two row
// parities (4 x 1024 doubles), symv_sym_gather_kernel adds them in a fixed order.
constexpr int SYP = 4 * SYT;   // doubles of `part` per tile: row sums [wc = 0, 1][1024], column sums [wr = 0, 1][1024]
constexpr int SYNB = 4;        // row buffers per wave

template <bool DIAG, bool EDGE>
__device__ __forceinline__ void symv_sym_tile_body(const int32_t* __restrict__ s32, int n, int nb,
                                                   const double* __restrict__ cm, const double mmean,
                                                   const double* __restrict__ x, double* __restrict__ part, int bi, int bj,
                                                   double* xs, double* ms, double (*rs)[SYT]) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (scalar: the row loops branch on it)
  const int wr = wave >> 1, wc = wave & 1;
  const int i0 = bi * SYT, j0 = bj * SYT;
  const int jw = j0 + 512 * wc + 4 * lane;   // the lane's 8 columns: jw + 256 q + {0..3}
  const int rows = EDGE ? min(SYT, n - i0) : SYT;
  for (int r = threadIdx.x; r < SYT; r += 256) {
    const bool in = !EDGE || i0 + r < n;
    xs[r] = in ? x[i0 + r] : 0.0;
    ms[r] = in ? cm[i0 + r] : 0.0;
  }
  double xj[8], mj[8], cacc[8];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int j = jw + 256 * q + e;
      const bool in = !EDGE || j < n;
      xj[4 * q + e] = in ? x[j] : 0.0;
      mj[4 * q + e] = in ? cm[j] : 0.0;
      cacc[4 * q + e] = 0.0;
    }
  __syncthreads();
  const int32_t* base = s32 + (int64_t)i0 * n + jw;
  auto load_row = [&](int r, int4 (&v)[2]) {
    const int32_t* rowp = base + (int64_t)r * n;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (EDGE) {
        const bool in = r < rows && jw + 256 * q < n;   // n % 4 == 0: a quad is whole or absent
        v[q] = in ? *reinterpret_cast<const int4*>(rowp + 256 * q) : make_int4(0, 0, 0, 0);
      } else {
        v[q] = *reinterpret_cast<const int4*>(rowp + 256 * q);
      }
    }
  };
  auto use_row = [&](int r, const int4 (&v)[2]) {
    const int i = i0 + r;
    const double xi = xs[r], mi = ms[r];
    double racc = 0.0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int sv[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = jw + 256 * q + e;
        const double d = (double)sv[e];
        double bij, bji;
        {
#pragma clang fp contract(off)
          bij = d - mi;
          bij = bij - mj[4 * q + e];
          bij = bij + mmean;
          bji = d - mj[4 * q + e];
          bji = bji - mi;
          bji = bji + mmean;
        }
        if (DIAG) {  // the tile's own upper triangle: the diagonal counts once (in the row sums)
          const bool in = !EDGE || j < n;
          racc += (j >= i && in) ? bij * xj[4 * q + e] : 0.0;
          cacc[4 * q + e] += (j > i && in) ? bji * xi : 0.0;
        } else if (EDGE) {
          // (columns >= n of the last block column: x and S were read as 0 but the centred entry is not 0 -> mask)
          racc += j < n ? bij * xj[4 * q + e] : 0.0;
          cacc[4 * q + e] += j < n ? bji * xi : 0.0;
        } else {
          racc += bij * xj[4 * q + e];
          cacc[4 * q + e] += bji * xi;
        }
      }
    }
    racc = wave_sum_to_lane63(racc);
    if (lane == 63) rs[wc][r] = racc;
    // one row at a time: left alone, the compiler computes the four row sums of the unrolled loop body first and keeps
    // their 32 converted entries alive for the column sums (183 registers, or spills) -- the column sums are pinned
    // here, and nothing moves across the barrier
#pragma unroll
    for (int c = 0; c < 8; ++c) asm volatile("" : "+v"(cacc[c]));
    __builtin_amdgcn_sched_barrier(0);
  };
  // rows wr, wr + 2, ..: SYNB buffers, each refilled as soon as its row is used
  int4 buf[SYNB][2];
  const int nr = rows > wr ? (rows - wr + 1) / 2 : 0;   // this wave's row count
#pragma unroll
  for (int b = 0; b < SYNB; ++b)
    if (b < nr) load_row(wr + 2 * b, buf[b]);
  int k = 0;
  for (; k + 2 * SYNB <= nr; k += SYNB) {   // every refill is a row of the tile: no conditions in the steady state
#pragma unroll
    for (int b = 0; b < SYNB; ++b) {
      use_row(wr + 2 * (k + b), buf[b]);
      load_row(wr + 2 * (k + b + SYNB), buf[b]);
    }
  }
#pragma unroll
  for (int b = 0; b < SYNB; ++b)   // the last < 2 SYNB rows
    if (k + b < nr) {
      use_row(wr + 2 * (k + b), buf[b]);
      if (k + b + SYNB < nr) load_row(wr + 2 * (k + b + SYNB), buf[b]);
    }
  k += SYNB;
#pragma unroll
  for (int b = 0; b < SYNB; ++b)
    if (k + b < nr) use_row(wr + 2 * (k + b), buf[b]);
  double* ptile = part + sym_tile_index(bi, bj, nb) * SYP;
  double* pcol = ptile + 2 * SYT + wr * SYT + 512 * wc + 4 * lane;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    *reinterpret_cast<double2*>(pcol + 256 * q) = make_double2(cacc[4 * q], cacc[4 * q + 1]);
    *reinterpret_cast<double2*>(pcol + 256 * q + 2) = make_double2(cacc[4 * q + 2], cacc[4 * q + 3]);
  }
  __syncthreads();
  for (int r = threadIdx.x; r < SYT; r += 256) {   // rows beyond N in the last block row: 0
    ptile[r] = r < rows ? rs[0][r] : 0.0;
    ptile[SYT + r] = r < rows ? rs[1][r] : 0.0;
  }
}

// One launch for all tiles (r05: as four launches the 97 diagonal tiles, the 97 tiles of the last block column and the corner
// ran one kind after the other behind the interior tiles -- 0.8 ms of a 4.2-ms mat-vec at N = 100,000).  The four bodies --
// interior tiles (no masks) and the tiles of the last block column / row when N is not a multiple of 1024 (EDGE), above the
// diagonal and on it -- are branches of one kernel (120 registers: four workgroups per CU); the few masked tiles come FIRST
// in the block order so that they run beside the interior ones instead of behind them.  nbi = block columns wholly inside N.
// blockIdx.x: [corner (nb - nbi)] [diagonal nbi] [last block column (nb - nbi) * (nb - 1)] [interior nbi (nbi - 1) / 2, row by row]
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(64)))  // (counts halves of the unified file: 128 registers)
void symv_sym_tiles_kernel(const int32_t* __restrict__ s32, int n, int nb, int nbi,
                           const double* __restrict__ cm, const double* __restrict__ stats,
                           const double* __restrict__ x, double* __restrict__ part) {
  __shared__ double xs[SYT], ms[SYT];  // 16 KiB: x_i and rowMean_i of the tile's rows
  __shared__ double rs[2][SYT];        // 16 KiB: the row sums over the two column halves
  const double mmean = stats[1];
  int t = blockIdx.x;
  const int edge = nb - nbi;           // 0 or 1
  if (t < edge) {
    symv_sym_tile_body<true, true>(s32, n, nb, cm, mmean, x, part, nb - 1, nb - 1, xs, ms, rs);
    return;
  }
  t -= edge;
  if (t < nbi) {
    symv_sym_tile_body<true, false>(s32, n, nb, cm, mmean, x, part, t, t, xs, ms, rs);
    return;
  }
  t -= nbi;
  if (t < edge * (nb - 1)) {
    symv_sym_tile_body<false, true>(s32, n, nb, cm, mmean, x, part, t, nb - 1, xs, ms, rs);
    return;
  }
  t -= edge * (nb - 1);
  int bi = 0;
  while (t >= nbi - 1 - bi) {
    t -= nbi - 1 - bi;
    ++bi;
  }
  symv_sym_tile_body<false, false>(s32, n, nb, cm, mmean, x, part, bi, bi + 1 + t, xs, ms, rs);
}

// y_i = over the tiles of block row bi, left to right, their two row sums (column half 0, then 1) + over the tiles of
// block column bi from the top down to the diagonal tile, their two column sums (row parity 0, then 1): a fixed order
__global__ __launch_bounds__(256) void symv_sym_gather_kernel(const double* __restrict__ part, int n, int nb, double* __restrict__ y) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int bi = i / SYT, r = i - bi * SYT;
  double acc = 0.0;
  for (int bj = bi; bj < nb; ++bj) {
    const double* t = part + sym_tile_index(bi, bj, nb) * SYP;
    acc += t[r] + t[SYT + r];
  }
  for (int bk = 0; bk <= bi; ++bk) {
    const double* t = part + sym_tile_index(bk, bi, nb) * SYP + 2 * SYT;
    acc += t[r] + t[SYT + r];
  }
  y[i] = acc;
}

// ---- large N: the exact row sums of S from its upper triangle (r05) ---------------------------------------------------------
// computePca's first pass (VariantsPca.scala:206: rowSums) read all N^2 entries, one workgroup per row with 4-byte loads:
// 11 ms of a 77-ms PCoA at N = 100,000.  S is symmetric and, once finalized, whole: an upper-triangular tile gives the row
// sums of its rows AND -- as column sums -- the contributions to the rows of its block column, in int64 (exact, as the
// reference's fold over integers below 2^53).  Same tiling, wave grid and prefetch as the mat-vec above; `part` is the
// mat-vec's workspace read as int64.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int64_t dpp_add_i64(int64_t v) {
  const int lo = (int)(uint32_t)v, hi = (int)(uint32_t)((uint64_t)v >> 32);
  const uint32_t l2 = (uint32_t)__builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xF, false);
  const uint32_t h2 = (uint32_t)__builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xF, false);
  return v + (int64_t)(((uint64_t)h2 << 32) | l2);
}
__device__ __forceinline__ int64_t wave_sum_i64_to_lane63(int64_t v) {
  v = dpp_add_i64<0xB1, 0xF>(v);
  v = dpp_add_i64<0x4E, 0xF>(v);
  v = dpp_add_i64<0x141, 0xF>(v);
  v = dpp_add_i64<0x140, 0xF>(v);
  v = dpp_add_i64<0x142, 0xA>(v);
  v = dpp_add_i64<0x143, 0xC>(v);
  return v;
}

__global__ __launch_bounds__(256) void rowsums_sym_tiles_kernel(const int32_t* __restrict__ s32, int n, int nb,
                                                                int64_t* __restrict__ part) {
  __shared__ int64_t rs[2][SYT];  // 16 KiB
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  int t = blockIdx.x, bi = 0;   // all tiles bi <= bj, row by row
  while (t >= nb - bi) {
    t -= nb - bi;
    ++bi;
  }
  const int bj = bi + t;
  const bool diag = bi == bj;
  const int i0 = bi * SYT, j0 = bj * SYT;
  const int jw = j0 + 512 * wc + 4 * lane;
  const int rows = min(SYT, n - i0);
  const bool full = j0 + SYT <= n;   // (rows < SYT only on the diagonal corner, where full is false as well)
  int64_t cacc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) cacc[c] = 0;
  const int32_t* base = s32 + (int64_t)i0 * n + jw;
  auto load_row = [&](int r, int4 (&v)[2]) {
    const int32_t* rowp = base + (int64_t)r * n;
#pragma unroll
    for (int q = 0; q < 2; ++q)
      v[q] = (full || (r < rows && jw + 256 * q < n)) ? *reinterpret_cast<const int4*>(rowp + 256 * q) : make_int4(0, 0, 0, 0);
  };
  auto use_row = [&](int r, const int4 (&v)[2]) {
    const int i = i0 + r;
    int64_t racc = 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int sv[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int j = jw + 256 * q + e;   // (entries at j >= n were loaded as 0)
        racc += (!diag || j >= i) ? sv[e] : 0;
        cacc[4 * q + e] += (!diag || j > i) ? sv[e] : 0;
      }
    }
    racc = wave_sum_i64_to_lane63(racc);
    if (lane == 63) rs[wc][r] = racc;
  };
  int4 buf[SYNB][2];
  const int nr = rows > wr ? (rows - wr + 1) / 2 : 0;
#pragma unroll
  for (int b = 0; b < SYNB; ++b)
    if (b < nr) load_row(wr + 2 * b, buf[b]);
  int k = 0;
  for (; k + 2 * SYNB <= nr; k += SYNB) {
#pragma unroll
    for (int b = 0; b < SYNB; ++b) {
      use_row(wr + 2 * (k + b), buf[b]);
      load_row(wr + 2 * (k + b + SYNB), buf[b]);
    }
  }
#pragma unroll
  for (int b = 0; b < SYNB; ++b)
    if (k + b < nr) {
      use_row(wr + 2 * (k + b), buf[b]);
      if (k + b + SYNB < nr) load_row(wr + 2 * (k + b + SYNB), buf[b]);
    }
  k += SYNB;
#pragma unroll
  for (int b = 0; b < SYNB; ++b)
    if (k + b < nr) use_row(wr + 2 * (k + b), buf[b]);
  int64_t* ptile = part + sym_tile_index(bi, bj, nb) * SYP;
  int64_t* pcol = ptile + 2 * SYT + wr * SYT + 512 * wc + 4 * lane;
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int e = 0; e < 4; ++e) pcol[256 * q + e] = cacc[4 * q + e];
  __syncthreads();
  for (int r = threadIdx.x; r < SYT; r += 256) {
    ptile[r] = r < rows ? rs[0][r] : 0;
    ptile[SYT + r] = r < rows ? rs[1][r] : 0;
  }
}

__global__ __launch_bounds__(256) void rowsums_sym_gather_kernel(const int64_t* __restrict__ part, int n, int nb,
                                                                 double* __restrict__ row_sums, int64_t* __restrict__ row_sums_i64) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int bi = i / SYT, r = i - bi * SYT;
  int64_t acc = 0;
  for (int bj = bi; bj < nb; ++bj) {
    const int64_t* t = part + sym_tile_index(bi, bj, nb) * SYP;
    acc += t[r] + t[SYT + r];
  }
  for (int bk = 0; bk <= bi; ++bk) {
    const int64_t* t = part + sym_tile_index(bk, bi, nb) * SYP + 2 * SYT;
    acc += t[r] + t[SYT + r];
  }
  row_sums_i64[i] = acc;
  row_sums[i] = (double)acc;
}

// doubles of workspace the symmetric form needs (0: not used at this N)
size_t symv_sym_workspace_doubles_impl(int n) {
  const int64_t nb = (n + SYT - 1) / SYT;
  return (size_t)(nb * (nb + 1) / 2) * (size_t)SYP;
}

void launch_symv(const EigWorkspace& ws, int n, const double* x, double* y, hipStream_t stream) {
  const unsigned rows4 = (unsigned)((n + 3) / 4);
  if (ws.a) {
    hipLaunchKernelGGL(symv_kernel, dim3(rows4), dim3(256), 0, stream, ws.a, n, x, y);
  } else if (ws.s64) {
    hipLaunchKernelGGL(symv_centered_kernel<true>, dim3(rows4), dim3(256), 0, stream, ws.s32, ws.s64, n, ws.colmean,
                       ws.stats, x, y);
  } else if (ws.sym_part && (n & 3) == 0) {
    const int nb = (n + SYT - 1) / SYT;
    const int nbi = n / SYT;   // block columns wholly inside N (nb or nb - 1)
    hipLaunchKernelGGL(symv_sym_tiles_kernel, dim3((unsigned)((int64_t)nb * (nb + 1) / 2)), dim3(256), 0, stream, ws.s32, n, nb, nbi,
                       ws.colmean, ws.stats, x, ws.sym_part);
    hipLaunchKernelGGL(symv_sym_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, ws.sym_part, n, nb, y);
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

constexpr int kMaxKrylov = 512;   // largest Krylov dimension (LDS arrays of the fused re-orthogonalisation kernels)

// Pass 1 of CGS2 on slice b = blockIdx.x of w (256 entries):  w[i] -= sum_p h[p] V[p][i],  and on the way out the dot
// products pass 2 needs, restricted to the slice:  part[b][q] = sum_{i in slice} V[q][i] w_new[i].  The next kernel
// adds the slices in a fixed order -- one launch less per Lanczos step than a second cgs_dots_kernel, no atomics.
__global__ __launch_bounds__(256) void cgs_update_dots_kernel(const double* __restrict__ v, int n, int count,
                                                              const double* __restrict__ h, double* __restrict__ w,
                                                              double* __restrict__ part) {
  __shared__ double red[4][kMaxKrylov + 8];
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool in = i < n;
  const int ic = in ? i : n - 1;
  double acc = 0.0;
  for (int p = 0; p < count; ++p) acc += h[p] * v[(int64_t)p * n + ic];
  const double wi = in ? w[ic] - acc : 0.0;
  if (in) w[i] = wi;
  for (int q = 0; q < count; ++q) {
    const double t = wave_sum(v[(int64_t)q * n + ic] * wi);
    if (lane == 0) red[wave][q] = t;
  }
  __syncthreads();
  for (int q = threadIdx.x; q < count; q += 256)
    part[(int64_t)blockIdx.x * count + q] = (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);
}

// Pass 2 on slice b:  h2[q] = sum_b part[b][q] (fixed order; every workgroup forms it for itself, workgroup 0 also
// publishes it),  w[i] -= sum_q h2[q] V[q][i],  pnorm[b] = sum_{i in slice} w_new[i]^2.
__global__ __launch_bounds__(256) void cgs_update_norm_kernel(const double* __restrict__ v, int n, int count,
                                                              const double* __restrict__ part, int nb,
                                                              double* __restrict__ w, double* __restrict__ h2,
                                                              double* __restrict__ pnorm) {
  __shared__ double hs[kMaxKrylov + 8];
  __shared__ double red[24];
  for (int q = threadIdx.x; q < count; q += 256) {
    double t = 0.0;
    for (int b = 0; b < nb; ++b) t += part[(int64_t)b * count + q];
    hs[q] = t;
    if (blockIdx.x == 0) h2[q] = t;
  }
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  double wi = 0.0;
  if (i < n) {
    double acc = 0.0;
    for (int p = 0; p < count; ++p) acc += hs[p] * v[(int64_t)p * n + i];
    wi = w[i] - acc;
    w[i] = wi;
  }
  const double t = block_sum(wi * wi, red);
  if (threadIdx.x == 0) pnorm[blockIdx.x] = t;
}

// alpha[j] = h1[j] + h2[j]; beta[j] = ||w|| (from the slices' partial sums); V[j+1] = w / beta[j]   (single workgroup)
__global__ __launch_bounds__(1024) void lanczos_finish_kernel(double* __restrict__ v, int n, int j,
                                                              const double* __restrict__ w,
                                                              const double* __restrict__ pnorm, int nb,
                                                              const double* __restrict__ h1,
                                                              const double* __restrict__ h2,
                                                              double* __restrict__ alpha, double* __restrict__ beta) {
  __shared__ double red[24];
  double part = 0.0;
  for (int b = threadIdx.x; b < nb; b += 1024) part += pnorm[b];
  const double nrm = sqrt(block_sum(part, red));
  const double rn = (nrm > 0.0) ? 1.0 / nrm : 0.0;
  double* vn = v + (int64_t)(j + 1) * n;
  for (int i = threadIdx.x; i < n; i += 1024) vn[i] = w[i] * rn;
  if (threadIdx.x == 0) {
    alpha[j] = h1[j] + h2[j];
    beta[j] = nrm;
  }
}

// The k Ritz values of largest magnitude among cand[0..count) -> sel[0..k), in the order the host derives from the
// same numbers afterwards (|.| descending, then value descending, then position): MLlib ranks by |lambda|.
__global__ __launch_bounds__(64) void ritz_select_kernel(const double* __restrict__ cand, int count, int k,
                                                         double* __restrict__ sel) {
  if (threadIdx.x != 0) return;
  // strict total order: a before b  <=>  |a| > |b|, or equal and a > b, or equal again and position(a) < position(b);
  // the t-th pick is the first element of that order behind the previous pick
  auto before = [&](int a, int b) {
    const double fa = fabs(cand[a]), fb = fabs(cand[b]);
    if (fa != fb) return fa > fb;
    if (cand[a] != cand[b]) return cand[a] > cand[b];
    return a < b;
  };
  int prev = -1;
  for (int t = 0; t < k && t < count; ++t) {
    int best = -1;
    for (int c = 0; c < count; ++c) {
      if (prev >= 0 && !before(prev, c)) continue;
      if (best < 0 || before(c, best)) best = c;
    }
    if (best < 0) break;
    sel[t] = cand[best];
    prev = best;
  }
}

// rec[0] = beta[m-1]; rec[1 + t] = last component of the t-th eigenvector of T_m (z[t][m-1])
__global__ __launch_bounds__(64) void ritz_collect_kernel(const double* __restrict__ beta, const double* __restrict__ z,
                                                          int m, int k, double* __restrict__ rec) {
  if (threadIdx.x == 0) rec[0] = beta[m - 1];
  for (int t = threadIdx.x; t < k; t += 64) rec[1 + t] = z[(int64_t)t * m + (m - 1)];
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

// ---- band Lanczos (r06): the fallback for CLUSTERED leading eigenvalues ------------------------------------------------------
// A single start vector sees one direction of a (near-)degenerate eigenspace; the second copy of a clustered eigenvalue
// only emerges at the rate of the tiny gap inside the cluster (lambda_1 ~ lambda_2 to 1e-9: never within 512 steps), and the
// dense O(N^3) solver the engine falls back to is disabled above N = 16,384 -- the reference's dgesdd decomposes any N it can
// hold (VariantsPca.scala:224-227; VERDICT r05 Missing 5).  Ruhe's band variant of block Lanczos with block width
// b = num_pc + 2: the basis starts with b orthonormal vectors and column j contributes v_{j+b} = orth(B v_j), one vector at a
// time, so every kernel of the single-vector iteration is re-used as is (full re-orthogonalisation by CGS2 against the whole
// basis).  Convergence is governed by the gap between the CLUSTER and the rest of the spectrum.  The projected matrix
// H = V^T B V (banded in exact arithmetic) is kept densely -- column j holds the CGS coefficients v_i^T B v_j, true projections
// whatever the basis looks like -- and its eigenproblem (<= 512 x 512) goes through the dense Householder + bisection + inverse
// iteration kernels of eig.hip.  A pair is returned only after its TRUE residual ||B u - theta u|| has passed.
//
// column j of H from the two CGS passes, the norm of what is left of w, and the normalised candidate vector into V[dst]
__global__ __launch_bounds__(1024) void band_finish_kernel(double* __restrict__ v, int n, int dst, const double* __restrict__ w,
                                                           const double* __restrict__ pnorm, int nb, const double* __restrict__ h1,
                                                           const double* __restrict__ h2, int cnt, double* __restrict__ hcol,
                                                           double* __restrict__ out2) {
  __shared__ double red[24];
  double part = 0.0;
  for (int b = threadIdx.x; b < nb; b += 1024) part += pnorm[b];
  const double nrm2 = block_sum(part, red);
  double hp = 0.0;
  for (int i = threadIdx.x; i < cnt; i += 1024) {
    const double h = h1[i] + h2[i];
    hcol[i] = h;
    hp += h * h;
  }
  const double h2sum = block_sum(hp, red);
  const double nrm = sqrt(nrm2);
  const double rn = (nrm > 0.0) ? 1.0 / nrm : 0.0;
  double* vn = v + (int64_t)dst * n;
  for (int i = threadIdx.x; i < n; i += 1024) vn[i] = w[i] * rn;
  if (threadIdx.x == 0) {
    hcol[cnt] = nrm;
    out2[0] = nrm;
    out2[1] = sqrt(h2sum + nrm2);   // ||B v_j||: the basis is orthonormal
  }
}

// after a thick restart the first p columns of H are the kept Ritz values on the diagonal (the Ritz vectors diagonalise H)
__global__ __launch_bounds__(64) void band_restart_h_kernel(double* __restrict__ hfull, int mcap, int p, const double* __restrict__ theta) {
  for (int t = threadIdx.x; t < p; t += 64) hfull[(int64_t)t * mcap + t] = theta[t];
}

// the leading J x J block of H as a dense symmetric matrix (upper triangle from the columns, mirrored)
__global__ __launch_bounds__(256) void band_gather_kernel(const double* __restrict__ hfull, int mcap, int J, double* __restrict__ a) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= J * J) return;
  const int i = t / J, j = t - i * J;
  a[t] = (i <= j) ? hfull[(int64_t)j * mcap + i] : hfull[(int64_t)i * mcap + j];
}

}  // namespace

size_t symv_sym_workspace_doubles(int32_t n) { return symv_sym_workspace_doubles_impl(n); }
hipError_t launch_row_sums_sym(const int32_t* s32, int32_t n, double* sym_part, double* row_sums, int64_t* row_sums_i64,
                               hipStream_t stream) {
  const int nb = (n + SYT - 1) / SYT;
  int64_t* part = reinterpret_cast<int64_t*>(sym_part);
  hipLaunchKernelGGL(rowsums_sym_tiles_kernel, dim3((unsigned)((int64_t)nb * (nb + 1) / 2)), dim3(256), 0, stream, s32, n, nb, part);
  hipLaunchKernelGGL(rowsums_sym_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, part, n, nb, row_sums,
                     row_sums_i64);
  return hipGetLastError();
}
void launch_centred_matvec(const EigWorkspace& ws, int32_t n, const double* x, double* y, hipStream_t stream) { launch_symv(ws, n, x, y, stream); }

size_t lanczos_workspace_doubles(int32_t n, int32_t k, int32_t mmax) {
  // V[(mmax+1)][n], w[n], bu[k][n], alpha[mmax], beta[mmax], h1[mmax+1], h2[mmax+1], the check record
  // (cand[2k+2], beta_m, ylast[k], res[k]), part[nb][mmax+1] + pnorm[nb] of the fused re-orthogonalisation
  const size_t nb = ((size_t)n + 255) / 256;
  // band fallback: H (mcap x mcap), its dense copy, d / e / tau / q (mcap each), w (2 mcap), scratch (6 mcap), the
  // eigenvectors of H (k x mcap), two scalars
  const size_t mcap = (size_t)mmax + 1;
  const size_t pmax = 2 * ((size_t)k + 2);   // band_pmax(k)
  const size_t band = 2 * mcap * mcap + 12 * mcap + 2 * pmax * mcap + 4 * pmax + 32;
  return (size_t)(mmax + 1) * n + (size_t)n + (size_t)k * n + 4 * (size_t)(mmax + 2) + 4 * (size_t)k + 16 +
         nb * (size_t)(mmax + 2) + band;
}

// Returns hipSuccess on a clean run; *converged tells whether ws.z[0..k) holds verified eigenvectors of
// B (unnormalised Ritz vectors; the caller normalises) and lam_sel_host[0..k) their eigenvalues
// (ordered by decreasing magnitude).  B (ws.a, or the implicit form when ws.a is null) is not modified.
namespace {
hipError_t lanczos_band(const EigWorkspace& ws, double* lz, int32_t n, int32_t k, int32_t mmax, double tol, double* lam_sel_host,
                        int* converged, int* band_steps_out, hipStream_t stream, const LanczosMatvec* mv);
inline int band_pmax(int k) { return 2 * (k + 2); }   // Ritz vectors a thick restart of the band iteration keeps, at most
hipError_t lanczos_single(const EigWorkspace& ws, double* lz, int32_t n, int32_t k, int32_t mmax, double tol,
                          double* lam_sel_host, int* converged, int* steps_out, hipStream_t stream, const LanczosMatvec* mv);
}

hipError_t lanczos_topk(const EigWorkspace& ws, double* lz, int32_t n, int32_t k, int32_t mmax, double tol,
                        double* lam_sel_host, int* converged, int* steps_out, hipStream_t stream, const LanczosMatvec* mv,
                        int* band_steps_out) {
  if (band_steps_out) *band_steps_out = 0;
  const bool band_only = debug_knobs().lanczos_band == 2 || ws.band_only;
  hipError_t e1 = band_only ? hipSuccess : lanczos_single(ws, lz, n, k, mmax, tol, lam_sel_host, converged, steps_out, stream, mv);
  if (band_only) *converged = 0;
  if (e1 != hipSuccess || *converged || debug_knobs().lanczos_band == 0) return e1;
  // clustered leading eigenvalues (or a spectrum the single vector resolves too slowly): the band iteration
  return lanczos_band(ws, lz, n, k, mmax, tol, lam_sel_host, converged, band_steps_out, stream, mv);
}

namespace {
hipError_t lanczos_single(const EigWorkspace& ws, double* lz, int32_t n, int32_t k, int32_t mmax, double tol,
                          double* lam_sel_host, int* converged, int* steps_out, hipStream_t stream, const LanczosMatvec* mv) {
  // mv: y = B v supplied by the caller (strip owners: B is tiled over GPUs and the product needs an all-gather the
  // engine knows nothing about); the stream is drained around the call
  auto matvec = [&](const double* v, double* y) -> hipError_t {
    if (!mv) {
      launch_symv(ws, n, v, y, stream);
      return hipSuccess;
    }
    hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    return (*mv)(v, y) == 0 ? hipSuccess : hipErrorUnknown;
  };
  *converged = 0;
  if (steps_out) *steps_out = 0;
  if (mmax > n) mmax = n;
  if (mmax > kMaxKrylov) mmax = kMaxKrylov;
  if (mmax < k + 2 || n < 8) return hipSuccess;  // tiny problems go to the dense path
  const unsigned nb = (unsigned)((n + 255) / 256);
  double* V = lz;
  double* w = V + (size_t)(mmax + 1) * n;
  double* bu = w + n;
  double* alpha = bu + (size_t)k * n;
  double* beta = alpha + (mmax + 2);
  double* h1 = beta + (mmax + 2);
  double* h2 = h1 + (mmax + 2);
  double* rec = h2 + (mmax + 2);                 // cand[2k+2] | beta_m | ylast[k] | res[k]
  double* part = rec + (4 * (size_t)k + 16);     // [nb][count] partial dot products of pass 2
  double* pnorm = part + (size_t)nb * (mmax + 1);

  EigWorkspace small = ws;  // T_m goes through the dense path's bisection / inverse iteration
  small.d = alpha;
  small.e = beta;

  hipLaunchKernelGGL(lanczos_init_kernel, dim3(1), dim3(1024), 0, stream, V, n);
  // Ritz pairs are examined at m = 12, 16, 20, 24, then every 8 steps up to 64, then every m / 2: a check costs about
  // three steps, and population structure converges early (configs[1] stand-in: estimate 1e-14 at m = 12, true
  // relative residual 3.7e-9)
  int next_check = 12;
  if (debug_knobs().lanczos_first_check > 0) next_check = std::max(4, debug_knobs().lanczos_first_check);
  // a check needs the k wanted Ritz values and one more for their gaps: T_m must have at least k + 2 rows (mmax does)
  if (next_check < k + 2) next_check = k + 2;
  if (next_check > mmax) next_check = mmax;
  std::vector<double> cand, pageable;
  std::vector<int32_t> idx;
  double est_prev = 0.0;
  for (int j = 0; j < mmax; ++j) {
    const double* vj = V + (size_t)j * n;
    const int cnt = j + 1;
    {
      const hipError_t em = matvec(vj, w);
      if (em != hipSuccess) return em;
    }
    // CGS2 in three launches: dots of pass 1 (one workgroup per basis vector), then per 256-entry slice of w the update
    // of pass 1 fused with the slice's share of the dots of pass 2, then the update of pass 2 fused with the slice's
    // share of ||w||^2.  (All of it in ONE workgroup was measured slower: 30 vs 24 us per step -- a single CU cannot
    // stream the ~2 MB of basis vectors a step touches as fast as many can.)
    hipLaunchKernelGGL(cgs_dots_kernel, dim3((unsigned)cnt), dim3(256), 0, stream, V, n, w, h1);
    hipLaunchKernelGGL(cgs_update_dots_kernel, dim3(nb), dim3(256), 0, stream, V, n, cnt, h1, w, part);
    hipLaunchKernelGGL(cgs_update_norm_kernel, dim3(nb), dim3(256), 0, stream, V, n, cnt, part, (int)nb, w, h2, pnorm);
    hipLaunchKernelGGL(lanczos_finish_kernel, dim3(1), dim3(1024), 0, stream, V, n, j, w, pnorm, (int)nb, h1, h2, alpha,
                       beta);
    const int m = j + 1;
    if (m != next_check && m != mmax) continue;
    next_check = (m < 24) ? m + 4 : (m < 64) ? m + 8 : m + m / 2;
    if (next_check > mmax) next_check = mmax;
    if (steps_out) *steps_out = m;

    // ---- the check, queued as a whole and read back ONCE (one host round trip instead of three): Ritz values of T_m
    // (k + 1 at each end: the k of largest magnitude are kept -- MLlib ranks by |lambda| -- and the extra ones bound
    // their gaps), their selection on the device, the eigenvectors y of T_m, the free estimate |beta_m y_last|, and --
    // speculatively, they are cheap -- the Ritz vectors u = V_m y and their TRUE residuals ||B u - theta u||.
    idx.clear();
    for (int t = 0; t <= k && t < m; ++t) idx.push_back(m - 1 - t);
    for (int t = 0; t <= k; ++t)
      if (t < m - 1 - k) idx.push_back(t);
    const int nc = (int)idx.size();
    double* rec_beta = rec + nc;          // beta_m, ylast[k]
    double* rec_res = rec_beta + 1 + k;   // res[k]
    hipError_t e = launch_bisect(small, m, idx.data(), nc, rec, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(ritz_select_kernel, dim3(1), dim3(64), 0, stream, rec, nc, k, ws.lam);
    if ((e = launch_inverse_iteration_dev(small, m, k, stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(ritz_collect_kernel, dim3(1), dim3(64), 0, stream, beta, ws.z, m, k, rec_beta);
    hipLaunchKernelGGL(ritz_kernel, dim3(nb, (unsigned)k), dim3(256), 0, stream, V, n, m, ws.z, bu);
    // ws.z <- u (ritz_kernel read y from ws.z, so it wrote to bu first)
    if ((e = hipMemcpyAsync(ws.z, bu, sizeof(double) * (size_t)k * n, hipMemcpyDeviceToDevice, stream)) != hipSuccess)
      return e;
    for (int t = 0; t < k; ++t) {
      const hipError_t em = matvec(ws.z + (size_t)t * n, bu + (size_t)t * n);
      if (em != hipSuccess) return em;
    }
    hipLaunchKernelGGL(residual_kernel, dim3((unsigned)k), dim3(1024), 0, stream, bu, ws.z, n, ws.lam, rec_res);
    const size_t rec_count = (size_t)nc + 1 + 2 * (size_t)k;
    double* hrec = ws.host_rec;
    if (!hrec || rec_count > ws.host_rec_cap) {
      pageable.resize(rec_count);
      hrec = pageable.data();
    }
    if ((e = hipMemcpyAsync(hrec, rec, sizeof(double) * rec_count, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
    cand.assign(hrec, hrec + nc);
    const double beta_m = hrec[nc];
    const double* ylast = hrec + nc + 1;
    const double* hres = ylast + k;

    std::vector<int> order(cand.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {   // == ritz_select_kernel's order
      const double fa = fabs(cand[a]), fb = fabs(cand[b]);
      if (fa != fb) return fa > fb;
      return cand[a] > cand[b];
    });
    double scale = 0.0;
    for (double c : cand) scale = fmax(scale, fabs(c));
    for (int t = 0; t < k; ++t) lam_sel_host[t] = cand[order[t]];
    if (!(scale > 0.0) || !std::isfinite(scale)) return hipSuccess;  // B == 0 or garbage: dense path decides
    // A Ritz pair is accepted when its residual is small against the spectrum AND against its own
    // gap (eigenvector error ~ residual / gap): target 1e-8, two orders inside the 1e-6 parity bar.
    std::vector<double> gap((size_t)k);
    for (int t = 0; t < k; ++t) {
      double g = DBL_MAX;
      for (size_t c2 = 0; c2 < cand.size(); ++c2)
        if ((int)c2 != order[t]) g = fmin(g, fabs(cand[c2] - lam_sel_host[t]));
      gap[(size_t)t] = g;
    }
    // ... unless the residual has reached what fp64 can deliver (1e-13 of the spectrum): inside a cluster of eigenvalues
    // the individual vectors are ill-conditioned for EVERY solver (r06: a pair of a 1e-9 cluster sat at a true residual of
    // 1e-16 from m = 144 on and was refused until m = 512)
    auto accept = [&](double r, int t) { return r <= tol * scale && (r <= 1e-8 * gap[(size_t)t] || r <= 1e-13 * scale); };
    bool ok = true;
    for (int t = 0; t < k; ++t) ok = ok && accept(fabs(beta_m * ylast[t]), t);
    const bool breakdown = beta_m <= 1e-14 * scale;  // invariant subspace: T_m holds exact eigenvalues
    bool verified = true;
    for (int t = 0; t < k; ++t) verified = verified && std::isfinite(hres[t]) && accept(hres[t] * 0.125, t);
    if (debug_knobs().lanczos_trace != 0) {
      std::fprintf(stderr, "[lanczos] m=%d beta_m=%.3e scale=%.6e", m, beta_m, scale);
      for (int t = 0; t < k; ++t)
        std::fprintf(stderr, "  theta%d=%.10e est=%.3e gap=%.3e true=%.3e", t, lam_sel_host[t], fabs(beta_m * ylast[t]),
                     gap[(size_t)t], hres[t]);
      std::fprintf(stderr, "  ok=%d verified=%d\n", (int)ok, (int)verified);
    }
    if (!ok && !breakdown) {   // the estimate says not yet: the speculative residual is not consulted
      if (m == mmax) return hipSuccess;
      // Stagnation: from m = 128 on a check comes every m / 2 steps; an estimate that has not improved fourfold since the
      // last one is a cluster the single vector cannot split (or a gap too small for this tolerance) -- the band iteration
      // takes over instead of the remaining steps up to mmax (3.4 ms each at N = 100,000).
      double est_now = 0.0;
      for (int t = 0; t < k; ++t) est_now = fmax(est_now, fabs(beta_m * ylast[t]) / scale);
      const bool stalled = m >= 128 && est_prev > 0.0 && est_now > 0.25 * est_prev;
      est_prev = est_now;
      if (stalled) return hipSuccess;
      continue;
    }
    if (verified) {
      *converged = 1;
      return hipGetLastError();
    }
    if (breakdown || m == mmax) return hipSuccess;  // not trustworthy: dense path
  }
  return hipGetLastError();
}

// The band iteration (see band_finish_kernel above).  Same contract as lanczos_single: *converged = 1 -> ws.z[0..k) holds the
// unnormalised Ritz vectors, lam_sel_host[0..k) their Ritz values by decreasing magnitude.  One host round trip per column (the
// norm of the new candidate decides whether it joins the basis): this is the fallback, not the fast path.
// THICK RESTARTS: the basis holds at most mmax vectors.  When it is full and the wanted pairs have not passed, the p = 2 b Ritz
// vectors of largest |theta| replace the processed part of the basis and the iteration goes on from the candidates that were
// still waiting: [u_1 .. u_p, v_J .. v_cnt) is orthonormal, the projected matrix on it is diag(theta) in its first p columns
// (Ritz vectors diagonalise H) and every later column is measured as it is processed -- so no spectrum, however slowly it
// converges, ends in "not converged" before a budget of ~40 basis fills is spent (ARPACK's scheme on the band process).
hipError_t lanczos_band(const EigWorkspace& ws, double* lz, int32_t n, int32_t k, int32_t mmax, double tol, double* lam_sel_host,
                        int* converged, int* band_steps_out, hipStream_t stream, const LanczosMatvec* mv) {
  auto matvec = [&](const double* v, double* y) -> hipError_t {
    if (!mv) {
      launch_symv(ws, n, v, y, stream);
      return hipSuccess;
    }
    hipError_t e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    return (*mv)(v, y) == 0 ? hipSuccess : hipErrorUnknown;
  };
  *converged = 0;
  if (mmax > n) mmax = n;
  if (mmax > kMaxKrylov) mmax = kMaxKrylov;
  const int ma = mmax;   // what the workspace was laid out for (lanczos_workspace_doubles; the same clamps as lanczos_single)
  if (debug_knobs().lanczos_band_mmax > 0 && debug_knobs().lanczos_band_mmax < mmax) mmax = debug_knobs().lanczos_band_mmax;   // tests: restarts
  const int bw0 = std::min<int>(k + 2, n);   // block width
  const int pk = std::min<int>(2 * bw0, band_pmax(k));   // Ritz vectors kept across a restart (the k wanted ones first)
  if (mmax < 4 * pk + 2 || n < 8) return hipSuccess;   // (a restart copies [v_J .. v_cnt) behind the kept Ritz vectors: no overlap from here)
  const unsigned nb = (unsigned)((n + 255) / 256);
  const int mcap = ma + 1;   // (layout of the workspace: independent of the mmax in use)
  double* V = lz;
  double* w = V + (size_t)(ma + 1) * n;
  double* bu = w + n;
  double* alpha = bu + (size_t)k * n;
  double* beta = alpha + (ma + 2);
  double* h1 = beta + (ma + 2);
  double* h2 = h1 + (ma + 2);
  double* rec0 = h2 + (ma + 2);
  double* part = rec0 + (4 * (size_t)k + 16);
  double* pnorm = part + (size_t)nb * (ma + 1);
  double* hfull = pnorm + nb;                       // [mcap][mcap], column j at hfull + j * mcap
  double* hdense = hfull + (size_t)mcap * mcap;     // J x J copy the dense solver works in
  double* hd = hdense + (size_t)mcap * mcap;
  double* he = hd + mcap;
  double* htau = he + mcap;
  double* hq = htau + mcap;
  double* hw = hq + mcap;                           // 2 mcap
  double* hscr = hw + 2 * (size_t)mcap;             // 6 mcap
  double* hout = hscr + 6 * (size_t)mcap;           // [pk][mcap] eigenvectors of H
  double* hz = hout + (size_t)band_pmax(k) * mcap;  // [pk][mcap] eigenvectors of the tridiagonal form (inverse iteration)
  double* rec = hz + (size_t)band_pmax(k) * mcap;   // candidates [2 pk + 2] | residuals [k]
  double* out2 = rec + (3 * (size_t)band_pmax(k) + 8);   // nrm, ||B v_j||
  double* hlam = out2 + 2;                          // [pk] selected Ritz values (device)

  EigWorkspace hs = ws;   // the projected problem goes through the dense solver's kernels
  hs.a = hdense; hs.d = hd; hs.e = he; hs.tau = htau; hs.q = hq; hs.w = hw; hs.scratch = hscr; hs.wy = nullptr;
  hs.z = hz; hs.lam = hlam;

  std::vector<double> pageable;
  auto read_back = [&](const double* src, size_t count, double** host) -> hipError_t {
    double* hrec = ws.host_rec;
    if (!hrec || count > ws.host_rec_cap) {
      pageable.resize(count);
      hrec = pageable.data();
    }
    hipError_t e = hipMemcpyAsync(hrec, src, sizeof(double) * count, hipMemcpyDeviceToHost, stream);
    if (e != hipSuccess) return e;
    if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
    *host = hrec;
    return hipSuccess;
  };
  hipError_t e = hipMemsetAsync(hfull, 0, sizeof(double) * (size_t)mcap * mcap, stream);   // entries no column ever measured are zeros
  if (e != hipSuccess) return e;

  // ---- the start block: bw0 pseudo-random vectors, orthonormalised by the same CGS2 kernels
  hipLaunchKernelGGL(lanczos_init_kernel, dim3(1), dim3(1024), 0, stream, V, n, 0u);
  int cnt = 1;   // basis vectors so far
  for (int t = 1; t < bw0; ++t) {
    hipLaunchKernelGGL(lanczos_init_kernel, dim3(1), dim3(1024), 0, stream, w, n, (uint32_t)t);
    hipLaunchKernelGGL(cgs_dots_kernel, dim3((unsigned)cnt), dim3(256), 0, stream, V, n, w, h1);
    hipLaunchKernelGGL(cgs_update_dots_kernel, dim3(nb), dim3(256), 0, stream, V, n, cnt, h1, w, part);
    hipLaunchKernelGGL(cgs_update_norm_kernel, dim3(nb), dim3(256), 0, stream, V, n, cnt, part, (int)nb, w, h2, pnorm);
    hipLaunchKernelGGL(lanczos_finish_kernel, dim3(1), dim3(1024), 0, stream, V, n, cnt - 1, w, pnorm, (int)nb, h1, h2, alpha, beta);
    cnt += 1;
  }
  double anorm = 0.0;
  int next_check = std::max(12, 3 * bw0);
  std::vector<double> cand;
  std::vector<int32_t> idx;
  const int cap = mmax - pk;          // basis vectors in use at most: pk slots stay free for the Ritz vectors of a restart
  const int64_t budget = 40LL * mmax; // columns (= mat-vecs) in all
  int64_t columns = 0;
  int restarts = 0;
  int j = 0;                          // next column to process
  for (;;) {
    // column j: w = B v_j, orthogonalised against the whole basis; what is left becomes v_cnt
    {
      const hipError_t em = matvec(V + (size_t)j * n, w);
      if (em != hipSuccess) return em;
    }
    columns += 1;
    hipLaunchKernelGGL(cgs_dots_kernel, dim3((unsigned)cnt), dim3(256), 0, stream, V, n, w, h1);
    hipLaunchKernelGGL(cgs_update_dots_kernel, dim3(nb), dim3(256), 0, stream, V, n, cnt, h1, w, part);
    hipLaunchKernelGGL(cgs_update_norm_kernel, dim3(nb), dim3(256), 0, stream, V, n, cnt, part, (int)nb, w, h2, pnorm);
    const bool room = cnt < cap;
    hipLaunchKernelGGL(band_finish_kernel, dim3(1), dim3(1024), 0, stream, V, n, cnt, w, pnorm, (int)nb, h1, h2, cnt,
                       hfull + (size_t)j * mcap, out2);   // (slot cnt <= cap is free either way: it is counted only with `room`)
    double* h2v = nullptr;
    if ((e = read_back(out2, 2, &h2v)) != hipSuccess) return e;
    const double nrm = h2v[0];
    anorm = fmax(anorm, h2v[1]);
    if (!std::isfinite(nrm) || !std::isfinite(anorm)) return hipSuccess;   // garbage: the dense path decides
    // a candidate that vanishes against the basis is deflated (the band narrows); none left = an invariant subspace
    if (room && nrm > 1e-9 * anorm) cnt += 1;
    const int J = j + 1;           // columns of H known so far
    j += 1;
    if (band_steps_out) *band_steps_out = (int)std::min<int64_t>(columns, 0x7fffffff);
    const bool exhausted = (J == cnt);        // every basis vector has been multiplied and nothing new came of it
    const bool full = !exhausted && cnt >= cap && J + bw0 >= cnt;   // the basis is full and its last candidates are next
    const bool out_of_budget = columns >= budget;
    const bool last = exhausted || out_of_budget;
    if (J != next_check && !last && !full) continue;
    if (J < k + 2 && !exhausted) continue;
    next_check = (J < 64) ? J + 8 : J + J / 4;

    // ---- Rayleigh-Ritz on span(v_0 .. v_{J-1}): dense eigenproblem of the leading J x J block of H
    if (J < 2) return hipSuccess;
    hipLaunchKernelGGL(band_gather_kernel, dim3((unsigned)((J * J + 255) / 256)), dim3(256), 0, stream, hfull, mcap, J, hdense);
    if ((e = launch_tridiagonalize(hs, J, stream)) != hipSuccess) return e;
    const int kk = std::min(k, J);
    const int pw = full ? std::min(pk, J - 1) : kk;   // eigenpairs of H wanted now: the k to test, or the pk a restart keeps
    idx.clear();
    for (int t = 0; t <= pw && t < J; ++t) idx.push_back(J - 1 - t);
    for (int t = 0; t <= pw; ++t)
      if (t < J - 1 - pw) idx.push_back(t);
    const int nc = (int)idx.size();
    double* rec_res = rec + nc;
    if ((e = launch_bisect(hs, J, idx.data(), nc, rec, stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(ritz_select_kernel, dim3(1), dim3(64), 0, stream, rec, nc, pw, hlam);
    if ((e = launch_inverse_iteration_dev(hs, J, pw, stream)) != hipSuccess) return e;
    if ((e = launch_backtransform(hs, J, pw, 0, 1, hout, stream)) != hipSuccess) return e;   // eigenvectors of H: hout[c * J + p]
    hipLaunchKernelGGL(ritz_kernel, dim3(nb, (unsigned)kk), dim3(256), 0, stream, V, n, J, hout, bu);
    if ((e = hipMemcpyAsync(ws.z, bu, sizeof(double) * (size_t)kk * n, hipMemcpyDeviceToDevice, stream)) != hipSuccess) return e;
    for (int t = 0; t < kk; ++t) {
      const hipError_t em = matvec(ws.z + (size_t)t * n, bu + (size_t)t * n);
      if (em != hipSuccess) return em;
    }
    hipLaunchKernelGGL(residual_kernel, dim3((unsigned)kk), dim3(1024), 0, stream, bu, ws.z, n, hlam, rec_res);
    double* hrec = nullptr;
    if ((e = read_back(rec, (size_t)nc + kk, &hrec)) != hipSuccess) return e;
    cand.assign(hrec, hrec + nc);
    const double* hres = hrec + nc;
    std::vector<int> order(cand.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {   // == ritz_select_kernel's order
      const double fa = fabs(cand[a]), fb = fabs(cand[b]);
      if (fa != fb) return fa > fb;
      return cand[a] > cand[b];
    });
    double scale = 0.0;
    for (double c : cand) scale = fmax(scale, fabs(c));
    if (!(scale > 0.0) || !std::isfinite(scale)) return hipSuccess;
    if (kk < k) return hipSuccess;   // fewer Ritz pairs than wanted (tiny invariant subspace): the dense path decides
    for (int t = 0; t < k; ++t) lam_sel_host[t] = cand[order[t]];
    // Acceptance, on the TRUE residual only.  While the iteration can still improve, a pair must also be resolved against its
    // own gap (eigenvector error ~ residual / gap <= 1e-8), as in the single-vector iteration -- unless the residual has reached
    // what fp64 can deliver (1e-13 of the spectrum): inside a cluster the individual vectors are ill-conditioned for EVERY
    // solver (LAPACK's dgesdd included), what is well defined is the pair's backward error.  At the last check the
    // backward-error bound alone decides.
    bool tight = true, loose = true;
    for (int t = 0; t < k; ++t) {
      double g = DBL_MAX;
      for (size_t c2 = 0; c2 < cand.size(); ++c2)
        if ((int)c2 != order[t]) g = fmin(g, fabs(cand[c2] - lam_sel_host[t]));
      const double r = hres[t];
      const bool fin = std::isfinite(r);
      loose = loose && fin && r * 0.125 <= tol * scale;
      tight = tight && fin && r * 0.125 <= tol * scale && (r * 0.125 <= 1e-8 * g || r <= 1e-13 * scale);
    }
    if (debug_knobs().lanczos_trace != 0) {
      std::fprintf(stderr, "[lanczos band] J=%d basis=%d restarts=%d columns=%lld scale=%.6e", J, cnt, restarts, (long long)columns, scale);
      for (int t = 0; t < k; ++t) std::fprintf(stderr, "  theta%d=%.12e true=%.3e", t, lam_sel_host[t], hres[t]);
      std::fprintf(stderr, "  tight=%d loose=%d last=%d full=%d\n", (int)tight, (int)loose, (int)last, (int)full);
    }
    if (tight || (last && loose)) {
      *converged = 1;
      return hipGetLastError();
    }
    if (last) return hipSuccess;
    if (!full) continue;

    // ---- thick restart: U = V_J Y (pw vectors) into the free slots, then [U, v_J .. v_cnt) becomes the basis
    hipLaunchKernelGGL(ritz_kernel, dim3(nb, (unsigned)pw), dim3(256), 0, stream, V, n, J, hout, V + (size_t)cnt * n);   // slots cnt .. cnt + pw - 1 <= mmax
    const int carry = cnt - J;   // candidates not yet multiplied
    if ((e = hipMemcpyAsync(V, V + (size_t)cnt * n, sizeof(double) * (size_t)pw * n, hipMemcpyDeviceToDevice, stream)) != hipSuccess) return e;
    if (carry > 0 &&   // (J > pw + carry: source and destination do not overlap)
        (e = hipMemcpyAsync(V + (size_t)pw * n, V + (size_t)J * n, sizeof(double) * (size_t)carry * n, hipMemcpyDeviceToDevice, stream)) != hipSuccess)
      return e;
    if ((e = hipMemsetAsync(hfull, 0, sizeof(double) * (size_t)mcap * mcap, stream)) != hipSuccess) return e;
    hipLaunchKernelGGL(band_restart_h_kernel, dim3(1), dim3(64), 0, stream, hfull, mcap, pw, hlam);
    cnt = pw + carry;
    j = pw;
    restarts += 1;
    next_check = j + std::max(8, 2 * bw0);
  }
}

}  // namespace

}  // namespace pcoa
