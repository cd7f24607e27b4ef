// Internal declarations shared by the HIP translation units of libpcoa_hip.so.
// gfx950 (MI355X / CDNA4) only: 64-wide wavefronts, MFMA, 160 KiB LDS per CU, 8 XCDs x 32 CUs.
#ifndef PCOA_INTERNAL_H_
#define PCOA_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <string>
#include <vector>

#include "pcoa.h"

namespace pcoa {

constexpr int kWave = 64;     // CDNA wavefront
constexpr int kNumXcd = 8;    // MI355X: block b is dispatched to XCD b % 8 (speed only, never correctness)

// ---- debug / experiment knobs -------------------------------------------------------------------
// Every environment variable the library looks at, read ONCE (first use) into this struct; nothing else calls getenv.
// None is needed for normal use; DESIGN_HISTORY.md "Environment hooks" documents them.
struct DebugKnobs {
  int gram_kernel = -1;            // PCOA_GRAM_KERNEL = auto | fp4 | i8 | f32  -> 0 | 3 | 2 | 1 (overrides the create flags)
  int64_t pack_chunk = 0;          // PCOA_DEBUG_PACK_CHUNK: variants per pre-pass launch (tests: multi-chunk paths)
  int64_t max_launch = 0;          // PCOA_DEBUG_MAX_LAUNCH: variants per contraction launch (tests: multi-launch paths)
  int64_t fold_threshold = 0;      // PCOA_DEBUG_FOLD_THRESHOLD: int32 -> int64 fold (tests: the fold / int64 all-reduce)
  int pipeline = -1;               // PCOA_PIPELINE = 0 | 1: force the two-stream fp32 pipeline off / on where it fits
  int explicit_center = 0;         // PCOA_EXPLICIT_CENTER: materialise B for the Lanczos path
  int lanczos_first_check = 0;     // PCOA_LANCZOS_FIRST_CHECK: Krylov dimension of the first Ritz check
  int lanczos_trace = 0;           // PCOA_DEBUG_LANCZOS: print the Ritz estimates
  int gram_cfg = 0;                // PCOA_GRAM_I8_CFG: contraction schedule (library built with -DPCOA_EXPERIMENTS only)
  int gram_splitk = 0;             // PCOA_GRAM_I8_SPLITK: split-K of the legacy launch
  int lockstep = -1;               // PCOA_GRAM_LOCKSTEP = 0 | 1: lock-step contraction launch off / on where it fits
  int guard = 0;                   // PCOA_DEBUG_GUARD = 1 | 2: every device buffer ends (1) / starts (2) at an unmapped page
  int operand = 0;                 // PCOA_OPERAND = fp4 | bits -> 1 | 2: operand of the binary-tile contraction (0 = default)
  int kbits_mode = -1;             // PCOA_KBITS_MODE = 0 | 2 | 4: launch form of the k-bits contraction (whole chip)
  int kbits_pipe_wgs = 0;          // PCOA_KBITS_PIPE_WGS: workgroups of the k-bits contraction beside the fp32 pre-pass
  int symv_sym_min_n = 0;          // PCOA_SYMV_SYM_MIN_N: smallest N whose Lanczos mat-vec reads only the upper triangle of S (default 16384)
  int csr_legacy = 0;              // PCOA_CSR_LEGACY = 1: pcoa_accumulate_calls through the host-validated r03 path
  int kbits_w4 = -1;               // PCOA_KBITS_W4 = 0 | 1 | 2: one-wave-per-SIMD contraction (gram_kbits_w4.inl): 0 never, 1 wherever the kernel has its CUs to itself (default), 2 also beside the ring pre-pass
  int kbits_w4_diag = -1;          // PCOA_KBITS_W4_DIAG: 0 = diagonal tiles as in r04a (one wave idles), 1..16 = wave roles on diagonal tiles with this cost (of 16) in the even split; default 11
  int kbits_coreside = -1;         // PCOA_KBITS_CORESIDE = 0 | 1: fp32 pipeline with pre-pass and contraction on the SAME CUs (ring pre-pass)
  int kbits_ring_prio = 0;         // PCOA_KBITS_RING_PRIO = 1: the ring pre-pass's waves at s_setprio 3 (harness knob)
  int kbits_ring_wgs = 0;          // PCOA_KBITS_RING_WGS: workgroups of the ring pre-pass beside a contraction (default 2 per CU)
  int u8_ring_wgs = 0;             // PCOA_U8_RING_WGS: workgroups of the uint8 ring pre-pass beside a contraction (default one per CU)
  int fork_lazy = 1;               // PCOA_FORK_LAZY = 0: a side stream waits on the ctx stream even when that is idle (r03 behaviour: a barrier packet in front of every pre-pass and contraction -- 2.26 vs 2.11 ms per fp32 step, profiles/r04w)
  int bits_pipeline = 0;           // PCOA_BITS_PIPELINE = 1: bitset tiles through the co-resident pipeline as in r03 / r04 (default: transpose and contraction in series, the contraction as the one-wave-per-SIMD kernel)
  int headstart_us = -1;           // PCOA_HEADSTART_US: the contraction's head start over the next pre-pass (default 10; 0 = none)
  int kbits_coreside_max_npad = 0; // PCOA_KBITS_CORESIDE_MAX_NPAD: largest padded sample count the co-resident pipeline is used for
  int lanczos_band_mmax = 0;       // PCOA_LANCZOS_BAND_MMAX: basis size of the band iteration (tests: forces thick restarts)
  int lanczos_band = 1;            // PCOA_LANCZOS_BAND = 0: no band-Lanczos fallback (r05 behaviour); 2: ONLY the band iteration (tests)
  int synth_tile = 0;              // PCOA_SYNTH_TILE = 1: pcoa_accumulate_synthetic through the fp32 staging tile + pre-pass (r05 path) instead of generating the k-bits operand directly
  int no_narrow = 0;               // PCOA_NO_NARROW = 1: an int64 S that fits int32 stays int64 and pcoa_gram_reduce_from always widens (r05 behaviour; tests of the int64 kernels)
};
const DebugKnobs& debug_knobs();

// ---- Gram kernels (gram_f32.hip / gram_packed.hip) ------------------------------------------------
struct GramLaunch {
  const float* x;        // device, [nv][ld] carrier multiplicities (0/1)
  int64_t ld;
  int64_t nv;            // variants in this launch (<= 2^24 so fp32 accumulators stay exact)
  int32_t n;             // samples
  int32_t* s32;          // device, [n][n] int32 partial (upper-triangular tiles only)
  int32_t* flag;         // device: bit 4 is raised when an fp32 accumulator leaves the exact range (|sum| >= 2^24)
  const float* zeros;    // device, >= 4 KiB of zeros (source for out-of-range rows)
  int num_cu;
  hipStream_t stream;
};
// Returns hipSuccess or the launch error.  *splitk_out (optional) receives the split-K factor used.
hipError_t launch_gram_f32(const GramLaunch& g, int* splitk_out);
// packed-operand path (gram_packed.hip): re-layout pre-pass into the k-blocked FP4 / int8 workspace, then the
// matrix-core contraction.
int64_t gram_packed_npad(int32_t n);
int64_t gram_packed_kb_pad_i8(int64_t nv);
size_t gram_packed_workspace_bytes(int32_t n, int64_t nv);
// flag[0]: bit 2 = a value that is not an integer in [0, 127]; flag[1] = max value seen (atomicMax): the carrier
// multiplicity bound the host sizes its int32 launches and folds by
hipError_t launch_pack_f32_i8(const float* x, int64_t ld, int64_t nv, int32_t n, int8_t* p, int32_t* flag,
                              hipStream_t stream);
hipError_t launch_pack_u8_i8(const uint8_t* x, int64_t ld, int64_t nv, int32_t n, int8_t* p, int32_t* flag,
                             hipStream_t stream);
hipError_t launch_densify_csr_i8(const int32_t* idx_dev, const int64_t* offs_dev, int64_t nv, int64_t offs_base,
                                 int8_t* p, int32_t n, int32_t* flag, hipStream_t stream);
// FP4 (MX E2M1) variant: fmt 1 = 32 variants per 16-byte lane slice, values must be exactly 0 / 1
int64_t gram_kb_pad(int64_t nv, int fmt);
hipError_t launch_densify_csr_fp4(const int32_t* idx_dev, const int64_t* offs_dev, int64_t nv, int64_t offs_base,
                                  int8_t* p, int32_t n, int32_t* flag, hipStream_t stream, int64_t nkb_out);
hipError_t launch_expand_bits_fp4(const uint32_t* bits, int64_t ld_words, int64_t nv, int32_t n, int8_t* p,
                                  hipStream_t stream, int64_t nkb_out = 0);
hipError_t launch_pack_fp4(const void* x, int is_u8, int64_t ld, int64_t nv, int32_t n, int8_t* p, int32_t* flag,
                           hipStream_t stream, int64_t nkb_out = 0);
// k-bits operand (gram_kbits.inl): K1[V/128][Npad][4 words], one BIT per genotype, expanded to FP4 in registers by the
// contraction.  nblk_out = blocks of 128 variants to write (the tail beyond nv is zero-filled).
hipError_t launch_pack_kbits(const void* x, int is_u8, int64_t ld, int64_t nv, int32_t n, int8_t* p, int32_t* flag,
                             hipStream_t stream, int64_t nblk_out);
hipError_t launch_transpose_bits_kbits(const uint32_t* bits, int64_t ld_words, int64_t nv, int32_t n, int8_t* p,
                                       hipStream_t stream, int64_t nblk_out);
hipError_t launch_densify_csr_kbits(const int32_t* idx_dev, const int64_t* offs_dev, int64_t nv, int64_t offs_base,
                                    int8_t* p, int32_t n, int32_t* flag, hipStream_t stream, int64_t nblk_out);
bool pack_fp4_ring_ok(const void* x, int64_t ld);
// persistent LDS-DMA-ring form of the fp32 -> k-bits pre-pass (needs pack_fp4_ring_ok): <= wgs workgroups; ring: 8 = 8 rows
// in flight per wave, default cache policy (what the library uses), 108 = nontemporal loads
hipError_t launch_pack_kbits_ring(const float* x, int64_t ld, int64_t nv, int32_t n, int8_t* p, int32_t* flag,
                                  hipStream_t stream, int64_t nblk_out, int wgs, int ring);
hipError_t launch_pack_fp4_ring(const float* x, int64_t ld, int64_t nv, int32_t n, int8_t* p, int32_t* flag,
                                hipStream_t stream, int64_t nkb_out, int wgs, int nt);
// the uint8 form: units of 128 variants x 1,024 samples, one wave per SIMD (ld % 8 == 0, 8-byte aligned base)
bool pack_u8_ring_ok(const void* x, int64_t ld);
hipError_t launch_pack_kbits_ring_u8(const uint8_t* x, int64_t ld, int64_t nv, int32_t n, int8_t* p, int32_t* flag,
                                     hipStream_t stream, int64_t nblk_out, int wgs, int ring);
// Strip owner (SURVEY 8e, N beyond one HBM): the launch computes S[:, col0 .. col0 + cols) -- every row block against the
// column blocks of the strip, BOTH triangles -- into a row-major [n][cols] matrix.  cols == 0: the ordinary symmetric job.
struct GramStrip {
  int col0 = 0, cols = 0;
  int cb0 = 0;  // first column block (filled in by the launcher)
};
// skip (optional, device): the launch does nothing when *skip != 0 -- the auto mode's device-side predicate (a pre-pass
// found a value other than 0 / 1 in the buffered tiles; the host learns it later and redoes them on the int8 kernel)
hipError_t launch_gram_packed(const int8_t* p, int fmt, int64_t nv, int32_t n, int32_t* s32, int num_cu,
                              hipStream_t stream, int* splitk_out, const int32_t* skip = nullptr,
                              GramStrip strip = GramStrip{});
// Lock-step launch: all tiles of `splitk` k-streams resident at once, one workgroup per CU for the whole launch.
// gram_lockstep_splitk: the k-stream count that fits `cus` CUs (8 XCDs), 0 if the shape does not fit.
// k-bits contraction; mode 0 = split-K launch, 2 = lock-step, 4 = even split of the (tile, stage) units over num_cu workgroups
hipError_t launch_gram_kbits(const int8_t* p, int64_t nv, int32_t n, int32_t* s32, int num_cu, hipStream_t stream, int mode,
                             const int32_t* skip = nullptr, GramStrip strip = GramStrip{});
// the same contraction with one wave per SIMD and 128 x 128 wave tiles (gram_kbits_w4.inl); same modes
hipError_t launch_gram_kbits_w4(const int8_t* p, int64_t nv, int32_t n, int32_t* s32, int num_cu, hipStream_t stream, int mode,
                                const int32_t* skip = nullptr, GramStrip strip = GramStrip{}, int wdiag = -1);
int gram_lockstep_splitk(int32_t n, int cus);
int gram_lockstep_workgroups(int32_t n, int splitk);
hipError_t launch_gram_packed_lockstep(const int8_t* p, int fmt, int64_t nv, int32_t n, int32_t* s32, int num_cu,
                                       hipStream_t stream, const int32_t* skip = nullptr);
hipError_t launch_gram_i8_packed(const int8_t* p, int64_t nv, int32_t n, int32_t* s32, int num_cu,
                                 hipStream_t stream, int* splitk_out);

// ---- auxiliary Gram kernels (gram_aux.hip) ----------------------------------------------------
hipError_t launch_densify_csr(const int32_t* idx_dev, const int64_t* offs_dev, int64_t v0, int64_t nv,
                              int64_t offs_base, float* x_dev, int64_t ld, int32_t n, int32_t* err_flag_dev,
                              hipStream_t stream);
// a one-wave spin (fp32 pipeline: head start for the contraction, pcoa_capi.hip fp4_setup)
hipError_t launch_delay_us(hipStream_t stream, int microseconds);
hipError_t launch_symmetrize_i32(int32_t* s32, int32_t n, hipStream_t stream);
hipError_t launch_fold_i32_to_i64(int32_t* s32, int64_t* s64, int64_t count, hipStream_t stream);
hipError_t launch_export_i64(const int32_t* s32, const int64_t* s64_or_null, int64_t* dst, int64_t count,
                             hipStream_t stream);
hipError_t launch_plink_bed_to_bits(const uint8_t* bed, int64_t row_bytes, int64_t nv, int32_t n, int64_t words, int ref_a1,
                                    uint32_t* bits, hipStream_t stream);
hipError_t launch_add_i64(int64_t* dst, const int64_t* src, int64_t count, hipStream_t stream);
hipError_t launch_add_i32(int32_t* dst, const int32_t* src, int64_t count, hipStream_t stream);
// s32[i] = (int32) s64[i] where it fits; flag[0] != 0 if some entry does not; ((int64*)flag)[1] = max |entry| (flag: 16 zeroed bytes)
hipError_t launch_narrow_i64_to_i32(const int64_t* s64, int32_t* s32, int64_t count, int32_t* flag, hipStream_t stream);
hipError_t launch_synth_fill_f32(uint64_t seed, const uint32_t* thresholds_dev, const int32_t* sample_pop_dev,
                                 int32_t n_pops, int64_t first_variant, int64_t nv, int32_t n, float* x_dev,
                                 int64_t ld, hipStream_t stream);

// the same genotypes straight into the k-bits operand (nblk_out blocks of 128 variants, the tail beyond nv zero; n_pops <= 64,
// nblk_out <= 65,535); thresholds_dev: [nv][n_pops] of exactly this range
hipError_t launch_synth_kbits(uint64_t seed, const uint32_t* thresholds_dev, const int32_t* sample_pop_dev, int32_t n_pops,
                              int64_t first_variant, int64_t nv, int32_t n, int32_t npad, int8_t* p, int64_t nblk_out,
                              hipStream_t stream);

// ---- centring (center.hip) --------------------------------------------------------------------
// s = s32 + (s64 ? s64 : 0).  row_sums[n] (fp64), stats[0] = matrix sum, stats[1] = matrix mean,
// nz[0] = #rows with sum > 0.  b = centred matrix fp64 [n][n].
// cm[j] = rowSums(j) / N (the division center_kernel performs per entry, done once)
hipError_t launch_col_means(const double* row_sums, int32_t n, double* cm, hipStream_t stream);
hipError_t launch_center(const int32_t* s32, const int64_t* s64_or_null, int32_t n, double* row_sums,
                         double* stats, int32_t* nz, double* b, hipStream_t stream, bool row_sums_done = false);

// ---- strip owner reductions (center.hip): S is [n][cols], column jj = sample col0 + jj
// ws: strip_ws_doubles(n, cols) doubles; the result (cols doubles) lands at ws[0 .. cols)
int64_t strip_ws_doubles(int32_t n, int32_t cols);
// column sums = the row sums of S for the strip's samples (S symmetric), exact integers as doubles
hipError_t launch_strip_col_sums(const int32_t* s32, const int64_t* s64_or_null, int32_t n, int32_t cols, double* ws,
                                 hipStream_t stream);
// y[jj] = sum_i B(col0 + jj, i) v[i],  B(j, i) = ((S(i, jj) - means[j]) - means[i]) + matrix_mean: row j of the centred
// matrix in the reference's operation order (VariantsPca.scala:216-221), from the strip's column (S is symmetric)
hipError_t launch_strip_matvec(const int32_t* s32, const int64_t* s64_or_null, int32_t n, int32_t col0, int32_t cols,
                               const double* v, const double* means, double matrix_mean, double* ws, hipStream_t stream);

// ---- symmetric eigensolver (eig.hip) ----------------------------------------------------------
struct EigWorkspace {
  double* a;        // [n][n] in: symmetric matrix B; out: reflector vectors in rows (row k, cols k+1..n-1)
  double* d;        // [n]   diagonal of T
  double* e;        // [n]   off-diagonal of T (n-1 used)
  double* tau;      // [n]   reflector scalars (n-2 used)
  double* q;        // [n]   A22 * v  (raw matvec of the current step)
  double* w;        // [n]   rank-2 update vector of the previous step
  double* lam;      // [2*kmax] candidate eigenvalues (k largest then k smallest)
  double* z;        // [kmax][n] eigenvectors of T, then of A (column c at z + c*n)
  double* scratch;  // [6][n] LU factors for inverse iteration
  int32_t* iscratch;// [2n + 64] pivots / eigenvalue indices
  double* wy;       // blocked back-transform: wy_workspace_doubles(n, kmax) doubles, or nullptr (serial form)
  double* host_rec = nullptr;   // pinned host memory for the Lanczos check record (host_rec_cap doubles), or nullptr
  size_t host_rec_cap = 0;
  // Implicit form of B for the Lanczos path: B(i,j) = ((S(i,j) - rowmean(i)) - colmean(j)) + mean is evaluated on
  // the fly from the integer S (the same expression, operation order and rounding as center_kernel, so the matvec sees
  // bit-identical entries) and the N x N fp64 matrix is never written.  a == nullptr then.
  const int32_t* s32;     // [n][n]
  const int64_t* s64;     // [n][n] or nullptr
  const double* colmean;  // [n] rowSums(j) / N
  const double* stats;    // stats[1] = matrixMean
  // large N: workspace of the symmetric mat-vec that reads only the upper triangle of S (symv_sym_workspace_doubles(n)
  // doubles: 1024 row sums + 1024 column sums per upper-triangular 1024 x 1024 tile), or nullptr: one wave per row
  double* sym_part = nullptr;
  bool band_only = false;   // PCOA_FLAG_EIG_BAND: skip the single-vector iteration (eigenvalues of multiplicity > 1)
};
size_t symv_sym_workspace_doubles(int32_t n);
// exact row sums of a finalized (symmetric) int32 S from its upper-triangular tiles: half the bytes of launch_center's row
// pass; sym_part = the mat-vec's workspace (symv_sym_workspace_doubles(n) doubles), n % 4 == 0
hipError_t launch_row_sums_sym(const int32_t* s32, int32_t n, double* sym_part, double* row_sums, int64_t* row_sums_i64,
                               hipStream_t stream);
void launch_centred_matvec(const EigWorkspace& ws, int32_t n, const double* x, double* y, hipStream_t stream);  // one y = B x (test hook)
hipError_t launch_tridiagonalize(const EigWorkspace& ws, int32_t n, hipStream_t stream);
// eigenvalues with ascending indices idx[0..count) of T -> lam_out[0..count) (device)
hipError_t launch_bisect(const EigWorkspace& ws, int32_t n, const int32_t* idx_host, int32_t count,
                         double* lam_out_dev, hipStream_t stream);
// eigenvectors of T for lam_sel[0..k) (host values) -> ws.z
hipError_t launch_inverse_iteration(const EigWorkspace& ws, int32_t n, const double* lam_sel_host, int32_t k,
                                    hipStream_t stream);
// the same with lam_sel already in ws.lam[0..k) on the device (one workgroup, vectors in sequence)
hipError_t launch_inverse_iteration_dev(const EigWorkspace& ws, int32_t n, int32_t k, hipStream_t stream);
// ws.z <- Q * ws.z (if apply_reflectors), normalise, sign-normalise (optional); out_dev[c*n + i] column-major
size_t wy_workspace_doubles(int32_t n, int32_t k);
hipError_t launch_backtransform(const EigWorkspace& ws, int32_t n, int32_t k, int sign_normalize,
                                int apply_reflectors, double* out_dev, hipStream_t stream);

// ---- Lanczos fast path (eig_lanczos.hip) ------------------------------------------------------
size_t lanczos_workspace_doubles(int32_t n, int32_t k, int32_t mmax);
// mv (optional): the caller's y = B v on device vectors (return 0); nullptr = the engine's own mat-vec on ws
typedef std::function<int(const double*, double*)> LanczosMatvec;
// band_steps_out (optional): basis vectors the band-Lanczos fallback built (0 = the single-vector iteration sufficed)
hipError_t lanczos_topk(const EigWorkspace& ws, double* lz, int32_t n, int32_t k, int32_t mmax, double tol,
                        double* lam_sel_host, int* converged, int* steps_out, hipStream_t stream,
                        const LanczosMatvec* mv = nullptr, int* band_steps_out = nullptr);

}  // namespace pcoa

#endif  // PCOA_INTERNAL_H_
