/*
 * pcoa.h -- C ABI of the MI355X-native PCoA engine (libpcoa_hip.so).
 *
 * Drop-in boundary for ONE path of googlegenomics/spark-examples: VariantsPcaDriver's
 *   getSimilarityMatrix -> computePca
 * (reference: src/main/scala/com/google/cloud/genomics/spark/examples/VariantsPca.scala, below
 * "VariantsPca.scala"; Python twin src/main/python/variants_pca.py, below "variants_pca.py").
 * The reference has no FFI; the seam is the method boundary of class VariantsPcaDriver
 * (VariantsPca.scala:81).  Every entry point below names the reference interface it replaces.
 * The JNI / ctypes bindings a maintainer would add are shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain pointers and sizes only; little-endian int32_t / int64_t / float / double;
 *   - every function returns 0 on success or a negative pcoa_status; it never throws or aborts;
 *     the message of the last failure is available from pcoa_last_error();
 *   - a pcoa_ctx owns all of its device memory and one HIP stream on ONE GPU; the caller owns every
 *     host buffer passed in; a ctx is used from one host thread at a time, different ctxs are
 *     independent (one per Spark task / per rank);
 *   - matrices are row-major unless stated; "device pointer" means HIP device memory on the ctx's GPU;
 *   - accumulate calls only queue work (on the ctx stream and, for device tiles -- fp32, uint8, bitsets --, on two side
 *     streams the ctx owns and orders against it: the pre-pass of one operand buffer runs while the previous one is contracted); the SYNCHRONISING calls are pcoa_sync, pcoa_gram_finalize, every read / load / export /
 *     import / all-reduce / compute call, pcoa_get_timings / pcoa_reset_timings and pcoa_set_stream.  A DEVICE input of an
 *     accumulate call must stay valid and unchanged until the next synchronising call returns: its pre-pass may still be
 *     running, and in the default (auto) mode a tile whose pre-pass met a carrier multiplicity is read a second time by
 *     the int8 path.  HOST inputs are consumed before the accumulate call returns.
 */
#ifndef PCOA_H_
#define PCOA_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCOA_VERSION_MAJOR 0
#define PCOA_VERSION_MINOR 6

typedef struct pcoa_ctx pcoa_ctx;

typedef enum pcoa_status {
  PCOA_OK = 0,
  PCOA_ERR_INVALID_ARG = -1,   /* null pointer, negative size, num_pc out of (0, N] ...          */
  PCOA_ERR_NO_DEVICE = -2,     /* no HIP device / ordinal out of range: the product has NO CPU fallback */
  PCOA_ERR_HIP = -3,           /* a HIP runtime call or kernel failed (message has the HIP error) */
  PCOA_ERR_OUT_OF_MEMORY = -4,
  PCOA_ERR_INDEX_RANGE = -5,   /* a callset index outside [0, N): the reference throws here
                                  (mapping(call.callsetId), VariantsPca.scala:59; Breeze bounds check :188) */
  PCOA_ERR_RCCL = -6,
  PCOA_ERR_NOT_CONVERGED = -7, /* no verified eigenpair (Lanczos-only mode, or N too large for the dense fallback) */
  PCOA_ERR_STATE = -8          /* the call does not apply to this kind of ctx (e.g. pcoa_compute on a strip owner) */
} pcoa_status;

/* flags for pcoa_create */
#define PCOA_FLAG_DEFAULT        0u     /* auto: MX-FP4 MFMA for binary (0/1) tiles, int8 MFMA for tiles that
                                           hold carrier multiplicities 2..127 (decided per operand-buffer
                                           generation, both exact; int32 launches and int64 folds are sized by the
                                           largest multiplicity met, so counts never wrap)                      */
#define PCOA_FLAG_GRAM_F32_MFMA  0x1u  /* fp32-MFMA Gram kernel (v_mfma_f32_32x32x2_f32): any small ints */
#define PCOA_FLAG_GRAM_I8_MFMA   0x2u  /* int8-MFMA Gram kernel only (v_mfma_i32_32x32x32_i8), values 0..127 */
#define PCOA_FLAG_GRAM_FP4_MFMA  0x4u  /* MX-FP4 Gram kernel only (v_mfma_f32_32x32x64_f8f6f4): a value
                                           other than 0 / 1 is an error                                        */
#define PCOA_FLAG_NO_SIGN_NORM   0x10u /* keep the eigensolver's native sign instead of sign-normalising */
#define PCOA_FLAG_EIG_HOUSEHOLDER 0x20u /* always use the dense Householder + bisection eigensolver        */
#define PCOA_FLAG_EIG_LANCZOS    0x40u /* Lanczos only: PCOA_ERR_NOT_CONVERGED instead of falling back      */
#define PCOA_FLAG_EIG_BAND       0x200u /* Lanczos as the BAND iteration from the start (block width num_pc + 2) instead of the
                                           single-vector one.  The default runs the single vector first (0.5 ms at N = 2504) and
                                           the band iteration only when that does not verify: clusters of eigenvalues down to a
                                           relative ~1e-12 are resolved either way, but an eigenvalue of EXACT multiplicity m > 1
                                           (a cohort with an exact symmetry: duplicated samples, mirrored populations) shows a
                                           single start vector one direction of its eigenspace only -- the pair that comes back
                                           is verified, yet a twin of the same eigenvalue can be missing from the top num_pc.
                                           This flag finds multiplicities up to num_pc + 2 (~3 ms instead of 0.5 at N = 2504). */
#define PCOA_FLAG_NO_PIPELINE    0x80u /* device tiles: pre-pass and contraction strictly one after the other on the ctx
                                           stream (the default runs the pre-pass of one fp32 / uint8 operand buffer beside
                                           the contraction of the previous one, on two side streams, for N = 1,025 .. 16,384;
                                           bitset tiles, whose pre-pass is a fifth of their contraction, run in series).
                                           The side streams want hardware queues of their own: HIP maps a process's streams
                                           onto a handful of queues per device, so with SEVERAL ctxs alive on ONE device the
                                           two kernels of a ctx can end up back to back (measured: uint8 tiles 1.62 instead of
                                           1.16 ms per step with a second, idle ctx alive) -- one ctx per device is the fast
                                           configuration */
#define PCOA_FLAG_OPERAND_FP4    0x100u /* binary tiles: keep the re-laid-out operand in HBM as MX-FP4 (4 bits per genotype,
                                           the r01 / r02 form) instead of the default k-bits form (1 bit per genotype,
                                           expanded to MX-FP4 in registers by the contraction).  Same MFMA, same S. */

/* Per-stage timings, filled by pcoa_get_timings(); times in seconds, measured with HIP events on
 * the ctx stream.  Counters are cumulative since pcoa_create / pcoa_reset_timings. */
typedef struct pcoa_timings {
  double gram_kernel_seconds;   /* sum of Gram-kernel launch durations                               */
  int64_t gram_kernel_launches;
  int64_t gram_variants;        /* variants pushed through the Gram kernels                           */
  double gram_flops;            /* algorithmic 2*V*N^2                                                */
  double gram_bytes;            /* algorithmic 4*V*N (X read once) + 4*N^2 per launch                 */
  double densify_seconds;       /* CSR -> dense tile kernels                                          */
  double synth_seconds;         /* synthetic genotype generation                                      */
  double finalize_seconds;      /* symmetrise + fold                                                  */
  double center_seconds;        /* row sums + double centring                                         */
  double tridiag_seconds;       /* Householder tridiagonalisation                                     */
  double eig_seconds;           /* tridiagonal eigenvalues + inverse iteration                        */
  double backtransform_seconds; /* reflector back-transform + normalisation                           */
  double compute_total_seconds; /* wall of the last pcoa_compute (centring..D2H)                      */
  int32_t gram_kernel_kind;     /* of the last launch: 1 = fp32 MFMA, 2 = int8 MFMA, 3 = MX-FP4 MFMA */
  int32_t operand_bits;         /* bits per genotype of that launch's operand in HBM: 32 (fp32 tile read in place), 8 (int8),
                                   4 (MX-FP4 operand), 1 (k-bits operand: expanded to MX-FP4 inside the contraction) */
  double pack_seconds;          /* fp32 -> k-blocked int8 pre-pass of the i8 path (sum of launches)   */
  int64_t pack_launches;
  double pack_bytes;            /* algorithmic bytes of the pre-pass: 4*V*N read + V*Npad written     */
  double lanczos_seconds;       /* Lanczos fast path of the eigensolver (sum over computes)           */
  int32_t eig_method;           /* of the last pcoa_compute: 1 = Lanczos (verified), 2 = Householder  */
  int32_t lanczos_steps;        /* Krylov dimension reached by the last pcoa_compute                  */
  int64_t fp4_fallbacks;        /* chunks the auto mode re-ran on the int8 kernel (non-binary values) */
  int64_t lockstep_launches;    /* contraction launches in the lock-step form (all tiles of a k-stream resident)  */
  int64_t pipeline_launches;    /* of those: launched on the contraction stream beside the next buffer's pre-pass
                                   (fp32 pipeline, DESIGN_HISTORY.md 4.1)                                                  */
  int32_t pipeline_pre_pass_cus;    /* CUs the pre-pass runs on while such a contraction does (0 = pipeline unavailable).  k-bits
                                       operand: all of them -- the ring pre-pass SHARES every CU with the contraction --, so
                                       pipeline_pre_pass_cus + pipeline_contraction_cus > the CU count means "co-resident" */
  int32_t pipeline_contraction_cus; /* CUs that contraction occupies (one workgroup each)                        */
  int64_t evensplit_launches;   /* contraction launches in the even-split form (k-bits operand: every workgroup an equal
                                   run of (tile, stage) units, one workgroup per CU)                                */
  /* ---- r04 (read them with pcoa_get_timings_sized; pcoa_get_timings stops in front of them) ---- */
  double csr_stage_seconds;     /* carrier lists: HOST seconds spent copying pageable arrays into pinned staging  */
  double csr_wait_seconds;      /* carrier lists: HOST seconds spent waiting for the device's check of a call     */
  int64_t csr_fast_chunks;      /* chunks (<= 8 M entries) scattered by the device-validated path                */
  int64_t csr_redo_chunks;      /* of those: redone on the int8 kernel because a list repeated a callset           */
  /* ---- r06 ---- */
  double allreduce_seconds;     /* pcoa_gram_allreduce_rccl: HIP-event time of the collective(s) on the ctx stream (agreement
                                   words + the S all-reduce), summed over calls                                            */
  int64_t allreduce_calls;
  int32_t comm_ranks;           /* ncclCommCount of the communicator of the last pcoa_gram_allreduce_rccl (0 = never called) */
  int32_t allreduce_int32;      /* 1 = that call reduced the int32 partial in place (4 N^2 bytes), 0 = the int64 branch     */
  int32_t matvec_form;          /* of the last pcoa_compute's Lanczos: 0 = one wave per row over all N^2 entries, 1 = upper-
                                   triangular tiles (N >= 16,384, no int64 part), 2 = materialised B                        */
  int32_t gram_i64_live;        /* 1 = S currently has an int64 part (counts beyond int32), 0 = S lives in the int32 matrix  */
  int64_t reduce_int32_calls;   /* pcoa_gram_reduce_from calls that took the int32 path (peer copy of 4 N^2 bytes)           */
  int64_t narrowed_to_int32;    /* times an int64 S handed in (import / load / int64 reductions) was found to fit int32 and
                                   moved back into the int32 matrix, which keeps the large-N upper-triangle forms available  */
  int32_t lanczos_block_steps;  /* band-Lanczos fallback (clustered leading eigenvalues): columns (= mat-vecs) it processed in
                                   the last pcoa_compute / pcoa_lanczos_with_matvec, thick restarts included; 0 = the
                                   single-vector iteration sufficed                                                          */
  int32_t reserved_r06;
} pcoa_timings;
#define PCOA_TIMINGS_R03_BYTES 192  /* offsetof(pcoa_timings, csr_stage_seconds): what pcoa_get_timings writes */

/* Synthetic genotype model (bench / tests only; not part of the reference).  Sample i belongs to
 * population p iff pop_offsets[p] <= i < pop_offsets[p+1].  Genotype X[v,i] = 1 iff
 * philox4x32-10(key = (seed_lo, seed_hi), counter = (v_lo, v_hi, i/4, 0))[i%4] < thresholds[v*n_pops + p].
 * Pure integer arithmetic => bit-identical on host and device and invariant to sharding. */
typedef struct pcoa_synth_params {
  uint64_t seed;
  int32_t n_pops;
  int32_t reserved;
  const int32_t* pop_offsets;   /* host, n_pops + 1 entries, pop_offsets[n_pops] == n_samples        */
  const uint32_t* thresholds;   /* host, [n_variants][n_pops] for the range being generated          */
} pcoa_synth_params;

/* ---- lifetime ------------------------------------------------------------------------------- */

/* Creates an engine for an N x N similarity matrix on GPU `device_ordinal`.
 * Replaces: the per-partition DenseMatrix.zeros[Int](size, size) (VariantsPca.scala:183-185) plus
 * the SparkContext the driver holds (VariantsPca.scala:83-85).  N = common.indexes.size. */
int pcoa_create(pcoa_ctx** out, int32_t n_samples, int32_t device_ordinal, uint32_t flags);

/* Strip owner (a cohort whose N x N matrix does not fit one HBM -- BASELINE configs[4], 250,000 samples,
 * S = 250 GB).  The ctx holds S[:, col0 .. col0 + cols) only -- every row, the strip's columns, BOTH triangles -- as a
 * row-major [N][cols] matrix; the strips of all owners tile S.  Every owner is fed ALL variants (the accumulate calls
 * below, unchanged; bitsets are 31 KB per variant at N = 250,000), so nothing is all-reduced: the exchange step is the
 * all-gather of an N-vector per Lanczos step (pcoa_strip_matvec).  pcoa_gram_read_i64 / _load_i64 / _export / _import
 * move [N][cols] matrices on a strip ctx, pcoa_gram_read_block_i64 takes (row, strip-relative column).
 * pcoa_center_read_f64 / pcoa_compute / pcoa_gram_allreduce_rccl are not available on it (PCOA_ERR_STATE).
 * Replaces: the same per-partition matrix of getSimilarityMatrix (VariantsPca.scala:183-189), column-sliced; the
 * reference itself stops at N^2 < 2^31 (DenseMatrix[Int]). */
int pcoa_create_strip(pcoa_ctx** out, int32_t n_samples, int32_t col0, int32_t cols, int32_t device_ordinal,
                      uint32_t flags);

/* computePca's eigensolver for a matrix the engine does not hold (r04): the engine's Lanczos iteration (Krylov basis and
 * re-orthogonalisation on the GPU, Ritz pairs by bisection + inverse iteration, a pair accepted only after its TRUE residual
 * passes) with the product y = B v supplied by the caller: fn(user, v_dev, y_dev) must leave B v in y_dev -- both are device
 * vectors of N doubles on the ctx's GPU -- with its own work complete when it returns, and return 0.  out_components:
 * [num_pc][N] unit, sign-normalised columns (as pcoa_compute); PCOA_ERR_NOT_CONVERGED if no verified pair is reached (there
 * is no dense fallback here).  Any ctx will do (a strip owner's: spark-examples_amd/strips.py runs computePca over strips
 * through this, every product one all-gather of an N-vector).  Replaces: the same MLlib call (VariantsPca.scala:224-227). */
typedef int (*pcoa_matvec_fn)(void* user, const double* v_dev, double* y_dev);
int pcoa_lanczos_with_matvec(pcoa_ctx* ctx, int32_t num_pc, pcoa_matvec_fn fn, void* user, double* out_components,
                             double* out_eigenvalues, int32_t* steps_out);

/* Returns 1 for a strip owner (and its column range), 0 for an ordinary ctx. */
int pcoa_strip_info(const pcoa_ctx* ctx, int32_t* col0_out, int32_t* cols_out);

/* out_cols[jj] = sum_i S(i, col0 + jj): the row sum of sample col0 + jj (S is symmetric), an exact integer.
 * Replaces: rowSums (VariantsPca.scala:206) for the strip's samples; the owners' vectors concatenate to rowSums. */
int pcoa_strip_col_sums(pcoa_ctx* ctx, double* out_cols);

/* y_out[jj] = sum_i B(col0 + jj, i) * v[i] with B(j, i) = ((S(j, i) - means[j]) - means[i]) + matrix_mean -- rows of the
 * double-centred matrix in the reference's operation order (VariantsPca.scala:216-221), evaluated on the fly from the
 * strip (B is never stored).  v, means (= rowSums / N, all N samples) and y_out are host arrays.  One call per owner and
 * Lanczos step; the owners' results concatenate (all-gather) to B v. */
int pcoa_strip_matvec(pcoa_ctx* ctx, const double* v, const double* means, double matrix_mean, double* y_out);

/* The same mat-vec for a host whose vectors already live on the GPU (e.g. torch tensors exchanged with an RCCL all-gather):
 * pcoa_strip_set_centering uploads means (host, N doubles) and matrix_mean ONCE per computePca -- they stay resident until
 * S changes -- and pcoa_strip_matvec_device takes v (N doubles) and y (cols doubles) as DEVICE pointers on the ctx's GPU.
 * It returns when y is complete; v must be complete when it is called (synchronise the producing stream first).
 * Nothing crosses PCIe per Lanczos step. */
int pcoa_strip_set_centering(pcoa_ctx* ctx, const double* means, double matrix_mean);
int pcoa_strip_matvec_device(pcoa_ctx* ctx, const double* v_dev, double* y_dev);

/* Replaces: VariantsPcaDriver.stop (VariantsPca.scala:283-285). */
void pcoa_destroy(pcoa_ctx* ctx);

/* Message of the last error on this ctx (or of the last failed pcoa_create when ctx == NULL).
 * Replaces: the JVM exception message. */
const char* pcoa_last_error(const pcoa_ctx* ctx);

/* Zeroes the accumulated similarity matrix (a fresh getSimilarityMatrix run). */
int pcoa_reset(pcoa_ctx* ctx);

int pcoa_n_samples(const pcoa_ctx* ctx);

/* Runs ctx work on a caller-provided hipStream_t (e.g. the host framework's current stream).
 * NULL restores the ctx-owned stream. */
int pcoa_set_stream(pcoa_ctx* ctx, void* hip_stream);

/* Blocks until all queued work of the ctx has finished. */
int pcoa_sync(pcoa_ctx* ctx);

/* Optional.  Allocates NOW what the first calls would otherwise allocate lazily: the operand buffers for accumulate calls
 * of up to `variants_per_call` variants (0 = skip) and the computePca workspace for `num_pc` components (0 = skip), so that
 * no device allocation happens inside a streaming or timed region (a first pcoa_compute is ~4x slower than a later one
 * without it).  Replaces: nothing in the reference -- the JVM allocates its per-partition matrix inside the task
 * (VariantsPca.scala:185); this is the executor warm-up a Spark host would run once per GPU. */
int pcoa_reserve(pcoa_ctx* ctx, int64_t variants_per_call, int32_t num_pc);

/* ---- Gram accumulation: getSimilarityMatrix -------------------------------------------------- */

/* The faithful boundary: exactly what RDD[Seq[Int]] carries (getCallsRdd, VariantsPca.scala:153-168).
 * Variant v's carriers are sample_idx[row_offsets[v] .. row_offsets[v+1]).  Host pointers.
 * Adds, for every variant and every ORDERED pair (c1, c2) of its carriers, 1 to S(c1, c2).
 * Replaces: the mapPartitions body of getSimilarityMatrix (VariantsPca.scala:184-189) and
 * sum_similarity (variants_pca.py:67-72).  Empty rows are legal (they add nothing; the reference
 * filters them at :166).  Repeated indices within a row count with multiplicity, as the
 * reference's double loop does (such a call runs on the int8 kernel; a call whose rows are sets --
 * everything a VCF yields -- goes to the MX-FP4 kernel; PCOA_FLAG_GRAM_FP4_MFMA turns a repeat into
 * PCOA_ERR_INVALID_ARG).  An index outside [0, N) -> PCOA_ERR_INDEX_RANGE, S unchanged by
 * that call's remaining chunks (reported at the next synchronising call at the latest). */
int pcoa_accumulate_calls(pcoa_ctx* ctx, const int32_t* sample_idx, const int64_t* row_offsets,
                          int64_t n_variants);

/* The same boundary with the caller saying where the arrays live (r04).  flags = 0 is pcoa_accumulate_calls: pageable host
 * arrays, consumed when the call returns.  What every form shares: the lists travel in chunks of <= 8 M entries through two
 * staging slots (H2D of chunk k+1 beside the scatter of chunk k), the DEVICE checks them (range, repeats), and a call is
 * only added to S once its check is clean -- PCOA_ERR_INDEX_RANGE leaves S as it was.
 *   PCOA_CALLS_HOST_PINNED  the arrays are page-locked (hipHostMalloc / hipHostRegister): the DMA engine reads them directly
 *                           (no copy into the library's pinned staging: ~10 GB/s per host thread against a 63 GB/s link).
 *   PCOA_CALLS_ASYNC        the arrays stay valid AND unmodified until the next synchronising call (pcoa_sync,
 *                           pcoa_gram_finalize, every read / compute / all-reduce / timing call): the call returns when
 *                           the work is queued; an error in its lists is reported by the call that next validates -- a later
 *                           accumulate call or that synchronising call -- and S is then unchanged by every call since the
 *                           last synchronising call that returned PCOA_OK.  A Spark host that builds batch k+1 while batch k
 *                           is on its way (VariantsPcaNative.scala) wants PINNED | ASYNC with two batch buffers.
 *   PCOA_CALLS_DEVICE_PTR   both arrays are device pointers (e.g. the output of a device-side VCF / BGEN decoder), read in
 *                           place; implies the lifetime rule of PCOA_CALLS_ASYNC, like every other device input.
 * Replaces: the same mapPartitions body (VariantsPca.scala:184-189); the partition iterator of :184 is what makes the
 * reference's boundary a stream of batches. */
#define PCOA_CALLS_DEVICE_PTR 1u
#define PCOA_CALLS_HOST_PINNED 2u
#define PCOA_CALLS_ASYNC 4u
int pcoa_accumulate_calls_ex(pcoa_ctx* ctx, const int32_t* sample_idx, const int64_t* row_offsets, int64_t n_variants,
                             uint32_t flags);

/* Dense boundary: a variants x samples tile, x[v*ld + i] = carrier multiplicity (0.0f / 1.0f for
 * well-formed input: extractCallInfo's hasVariation, VariantsPca.scala:56-60).  `x` is a host
 * pointer (is_device_ptr = 0; staged through pinned memory) or a device pointer (is_device_ptr = 1;
 * read in place, must stay valid until the next synchronising call).  ld >= N.
 * Replaces: the same mapPartitions body, with the RDD partition materialised as a matrix tile
 * (BASELINE.json configs[1]: "2,504 samples x 1M variants fp32").
 * Rates: a tile of 0 / 1 values runs the pipelined k-bits / MX-FP4 path.  A tile that holds carrier multiplicities
 * 2..127 (not produced by the reference's call extraction: hasVariation is a boolean) takes the int8 path, which is NOT
 * pipelined -- its pre-pass (one byte per genotype, 8x the operand bytes) and its contraction (int8 MFMA: half the MX-FP4
 * rate) run in series on the ctx stream -- at about half the rate of a binary tile (225 vs 480 M variants/s for fp32
 * tiles at N = 2504), and in the auto mode the tile is read twice (the binary pre-pass finds the multiplicity first).
 * Any ld >= N that keeps rows 16-byte aligned (ld % 4 == 0) takes the fast pre-pass; the pitch needs no padding to
 * 128-byte lines (measured: profiles/r04y_ring_pitch_sweep.txt). */
int pcoa_accumulate_dense_f32(pcoa_ctx* ctx, const float* x, int64_t n_variants, int64_t ld,
                              int is_device_ptr);

/* Same boundary with one byte per genotype (values 0..127): the production format -- a 4x cheaper re-layout pass than the
 * fp32 tile (SURVEY.md 8f "bit-packed / int8 Gram path").  Kernel choice as for the fp32 tile: in the default (auto) mode a
 * tile of 0 / 1 bytes goes to the MX-FP4 matrix cores (pcoa_timings.gram_kernel_kind == 3), a tile with carrier
 * multiplicities 2..127 to the int8 ones (== 2); the create flags force either.  Host or device pointer as above. */
int pcoa_accumulate_dense_u8(pcoa_ctx* ctx, const uint8_t* x, int64_t n_variants, int64_t ld, int is_device_ptr);

/* Bit-packed variants x samples tile: row v is the carrier BITSET of variant v, 1 bit per genotype; sample i is
 * bit (i & 31) of the little-endian word bits[v * ld_words + (i >> 5)], ld_words >= ceil(N / 32); bits of samples
 * >= N are ignored.  313 B per variant at N = 2504 (32x less than the fp32 tile: BASELINE configs[2]'s 40 M variants
 * are 12.5 GB).  Host or device pointer.  A bitset cannot repeat a callset, so the tile is binary by construction and
 * always runs on the MX-FP4 kernel (not available with PCOA_FLAG_GRAM_F32_MFMA).
 * Replaces: the same RDD[Seq[Int]] rows as pcoa_accumulate_calls (getCallsRdd, VariantsPca.scala:153-168; the
 * hasVariation indicator of extractCallInfo, :56-60), one bit per (variant, callset) instead of a list of indices;
 * SURVEY 8(d) "1-bit-packed twin", 8(f) rank 2. */
int pcoa_accumulate_bits(pcoa_ctx* ctx, const uint32_t* bits, int64_t n_variants, int64_t ld_words,
                         int is_device_ptr);

/* PLINK 1 binary genotypes as they lie in a variant-major .bed file (r04): bed_rows[v * row_bytes + s / 4] holds sample s of
 * variant v in bits 2 (s % 4): 00 homozygous A1, 01 missing, 10 heterozygous, 11 homozygous A2; row_bytes >= ceil(N / 4) (the
 * file's own pitch is exactly that; the three magic bytes are the caller's to skip).  A sample is a carrier of the variant --
 * hasVariation, VariantsPca.scala:56-60 -- iff its code is 10 or the homozygous NON-reference one: 00 when A2 is the reference
 * allele (plink --keep-allele-order / plink2 --make-bed; ref_is_a1 = 0), 11 when A1 is (ref_is_a1 = 1); a missing call
 * carries nothing.  The decode runs on the device (a host only reads the file: 626 B per variant at N = 2504), then the
 * bitset path of pcoa_accumulate_bits.  is_device_ptr: 0 host rows (consumed when the call returns), 1 device rows (lifetime
 * rule of every device input), PCOA_BED_HOST_ASYNC (r05) host rows in PAGE-LOCKED memory (pcoa_host_alloc_pinned) that the
 * call only QUEUES: it returns while the copy may still be running, so a host that rotates a few blocks keeps the link busy
 * while it reads the next block.  The rows of such a call must stay unmodified until the SECOND later call of this
 * function (with n_variants > 0) on the same ctx has returned (host rows travel through two device slots: a call that takes a slot first waits
 * for the decode of the call that used it last) or until any synchronising call (pcoa_sync, pcoa_gram_finalize, ..).
 * Replaces: the same RDD[Seq[Int]] rows (getCallsRdd, VariantsPca.scala:153-168) for a cohort stored as a PLINK fileset. */
#define PCOA_BED_HOST_ASYNC 2
int pcoa_accumulate_plink_bed(pcoa_ctx* ctx, const uint8_t* bed_rows, int64_t n_variants, int64_t row_bytes, int ref_is_a1,
                              int is_device_ptr);

/* Generates variants [first_variant, first_variant + n_variants) of the synthetic model directly
 * in HBM and accumulates them (no host tile).  params->thresholds covers exactly that range.
 * r06: on the default engine the genotypes are written straight into the 1-bit operand (no fp32 staging tile, no pre-pass,
 * no host synchronisation per chunk); the call returns when the thresholds have been copied, the rest is queued. */
int pcoa_accumulate_synthetic(pcoa_ctx* ctx, const pcoa_synth_params* params, int64_t first_variant,
                              int64_t n_variants);

/* Fills a caller-owned DEVICE buffer x_dev[n_variants][ld] with the synthetic genotypes (0.0f/1.0f)
 * without accumulating -- used by bench.py to make the input resident before the timed region. */
int pcoa_synth_fill_f32(pcoa_ctx* ctx, const pcoa_synth_params* params, int64_t first_variant,
                        int64_t n_variants, float* x_dev, int64_t ld);

/* Completes the local accumulation: contracts what the accumulate calls have only packed so far (binary tiles
 * are re-laid out into a packed operand buffer (1 bit per genotype by default) when they arrive and contracted when the
 * buffer is full: 2^20 variants where two buffers alternate, never more than 2^22,
 * or here -- every reader of S below does the same), mirrors the computed triangle, folds int32 partials.
 * After it the full symmetric S of THIS ctx is readable.  Accumulation may continue afterwards
 * (S is additive: a natural checkpoint/resume point).  Inputs of the accumulate calls are consumed by their
 * pre-pass: a device pointer has to stay valid until the next synchronising call, not until the contraction.
 * Replaces (single GPU): reduceByKey(_ + _) (VariantsPca.scala:190). */
int pcoa_gram_finalize(pcoa_ctx* ctx);

/* Cross-GPU sum of the finalized S over all ranks of an RCCL communicator (ncclComm_t passed as
 * void*), in place on every rank, on the ctx stream.  One rank per GPU / per process.
 * Replaces (multi GPU): reduceByKey(_ + _, numReducePartitions) (VariantsPca.scala:190). */
int pcoa_gram_allreduce_rccl(pcoa_ctx* ctx, void* nccl_comm);

/* RCCL bootstrap helpers for hosts without their own (e.g. the Scala/JNI host): rank 0 obtains a
 * 128-byte unique id, ships it to the other ranks by its own means (Spark broadcast), then every
 * rank calls pcoa_comm_init.  *comm_out is an ncclComm_t. */
int pcoa_comm_unique_id(uint8_t out_id[128]);
/* Which RCCL the three calls above and pcoa_gram_allreduce_rccl are bound to.  libpcoa_hip.so does not link librccl: at
 * the first communicator call it binds the RCCL image the process has already mapped (a PyTorch process carries its own,
 * torch/lib/librccl.so -- one collective runtime per process, not two), else librccl.so.1 from the library's RUNPATH
 * (/opt/rocm/lib: the Scala / JNI host's case).  path_out receives the image's path, *version_out ncclGetVersion's code;
 * either may be NULL.  PCOA_ERR_RCCL if no RCCL can be found. */
int pcoa_comm_runtime(char* path_out, int32_t path_cap, int32_t* version_out);
int pcoa_comm_init(pcoa_ctx* ctx, const uint8_t id[128], int32_t rank, int32_t n_ranks,
                   void** comm_out);
int pcoa_comm_destroy(void* nccl_comm);
/* Number of ranks of a communicator (ncclCommCount) -- what a bench line or a host log prints to show that the collective
 * really spans the GPUs it was launched on.  *count_out = ranks. */
int pcoa_comm_count(void* nccl_comm, int32_t* count_out);

/* Page-locked host memory for input blocks a host fills and hands to the accumulate calls (hipHostMalloc): the DMA engine
 * reads it at link speed, pageable memory goes through the runtime's staging copy at a third of that.  Process-wide, not tied
 * to a ctx; at least one ctx (i.e. a HIP device) must exist.  (A JVM host passes such a block as a direct ByteBuffer.) */
int pcoa_host_alloc_pinned(size_t bytes, void** out);
int pcoa_host_free_pinned(void* p);

/* dst.S += src.S for two engines of the SAME process (r04): what a host with one ctx per GPU -- k host threads, no
 * collective runtime -- calls after the accumulation (host/variants_pca_driver --gpus k; a Spark executor with several GPUs).
 * Both ctxs are synchronised; src's total crosses by hipMemcpyPeerAsync (xGMI on a node; the ctxs may also share a device).
 * r06: when neither engine has an int64 part and the summed variant weights stay below 2^31 (the test pcoa_gram_allreduce_rccl
 * applies) the int32 matrices are added in place -- 4 N^2 bytes cross, nothing is widened, and the large-N upper-triangle
 * forms of computePca stay available on dst; otherwise src's total leaves as int64 and is added into dst's int64 matrix.
 * src is unchanged.  Integer sums: order and grouping of the reductions do not matter.
 * Replaces: reduceByKey(_ + _, conf.numReducePartitions()) (VariantsPca.scala:190, GenomicsConf.scala:42-45) inside one JVM. */
int pcoa_gram_reduce_from(pcoa_ctx* dst, pcoa_ctx* src);

/* Host-driven reduction alternative (bench.py uses torch.distributed, whose "nccl" backend is RCCL):
 * export copies the finalized S as int64 [N][N] into a DEVICE buffer; import replaces S with the
 * (reduced) contents of a DEVICE buffer. */
int pcoa_gram_export_device_i64(pcoa_ctx* ctx, int64_t* dst_dev);
int pcoa_gram_import_device_i64(pcoa_ctx* ctx, const int64_t* src_dev);

/* Copies the finalized S to host as int64 [N][N] (all N^2 entries, zeros included, as
 * matrix.iterator emits them, VariantsPca.scala:189).  For parity tests and checkpoints. */
int pcoa_gram_read_i64(pcoa_ctx* ctx, int64_t* out_nxn);

/* Copies the block S[row0 .. row0+rows) x [col0 .. col0+cols) of the finalized S to host (row-major,
 * rows x cols int64).  For spot checks and sliced checkpoints when N^2 entries are too many to move. */
int pcoa_gram_read_block_i64(pcoa_ctx* ctx, int32_t row0, int32_t col0, int32_t rows, int32_t cols, int64_t* out);

/* Loads S from host int64 [N][N] (resume from a checkpoint, or enter at computePca with matrix
 * entries produced elsewhere: computePca(matrixEntries), VariantsPca.scala:198). */
int pcoa_gram_load_i64(pcoa_ctx* ctx, const int64_t* in_nxn);

/* ---- computePca ------------------------------------------------------------------------------ */

/* Runs row sums + double-centring only and copies B (fp64 [N][N]) and the row sums to host.
 * Either output may be NULL.  Replaces: computePca lines VariantsPca.scala:206-223
 * (center_matrix, variants_pca.py:84-121).  For parity tests of that stage. */
int pcoa_center_read_f64(pcoa_ctx* ctx, double* out_b_nxn, double* out_row_sums, int32_t* out_nonzero_rows,
                         double* out_matrix_mean);

/* computePca (VariantsPca.scala:198-231; perform_pca variants_pca.py:123-152) on the finalized S:
 * centring, then the num_pc principal components of B.
 *   out_components : N x num_pc COLUMN-major, i.e. the layout of pca.toArray (VariantsPca.scala:227):
 *                    component c of sample i is out_components[i + c*N]; unit 2-norm columns;
 *                    sign-normalised (largest-magnitude entry positive, ties -> lowest index)
 *                    unless PCOA_FLAG_NO_SIGN_NORM.
 *   out_eigenvalues: num_pc eigenvalues of B, ordered by decreasing magnitude -- the order of the
 *                    singular values of Cov that MLlib's SVD sorts by (s = lambda^2/(N-1)). May be NULL.
 *   out_nonzero_rows: rowSums.filter(_ > 0).size (VariantsPca.scala:207). May be NULL.
 * 0 < num_pc <= N, else PCOA_ERR_INVALID_ARG (MLlib: require(k > 0 && k <= n)).
 * Solver: Lanczos on the centred matrix (never materialised), a pair returned only after its TRUE residual
 * ||B u - theta u|| passed on the device; if the single-vector iteration does not verify (clustered leading eigenvalues,
 * slow spectra) the band iteration with thick restarts takes over (r06), then -- up to N = 16,384 -- the dense Householder
 * solver.  PCOA_ERR_NOT_CONVERGED is what is left when all of them fail (pcoa_timings.eig_method / lanczos_block_steps
 * tell which one answered). */
int pcoa_compute(pcoa_ctx* ctx, int32_t num_pc, double* out_components, double* out_eigenvalues,
                 int32_t* out_nonzero_rows);

/* ---- instrumentation ------------------------------------------------------------------------- */

/* Synchronises and fills *out.  Replaces: reportIoStats' role of printing what was processed
 * (VariantsPca.scala:48,281). */
int pcoa_get_timings(pcoa_ctx* ctx, pcoa_timings* out);
/* The struct grows at its end from release to release: pcoa_get_timings writes the r03 layout (PCOA_TIMINGS_R03_BYTES) and
 * nothing behind it; pass sizeof(pcoa_timings) of the header you compiled against here to receive the newer fields too. */
int pcoa_get_timings_sized(pcoa_ctx* ctx, pcoa_timings* out, size_t out_size);
int pcoa_reset_timings(pcoa_ctx* ctx);

/* ---- test hooks (not part of the reference-facing boundary) ---------------------------------------------------------------
 * Device memory from the library's own allocator.  With the environment variable PCOA_DEBUG_GUARD=1 (=2) every device
 * buffer of the library -- and these -- is a virtual range of its own whose END (START) lies against a page that is
 * never mapped: a kernel that reads or writes one element too far faults at once ("Memory access fault by GPU") instead
 * of touching a neighbouring allocation.  The GPU test suite runs its fuzz and parity sweeps in that mode with the
 * input tiles allocated here (tests/test_gpu_guard.py).  pcoa_debug_guard_mode returns the mode in effect (0 = off). */
int pcoa_debug_alloc(int32_t device_ordinal, size_t bytes, void** out);
/* One y = B x of the centred matrix of the current S (host vectors of N doubles) with either form of the eigensolver's mat-vec:
 * 0 = one wave per row over all N^2 entries, 1 = upper-triangular 1024 x 1024 tiles, each entry read once and used for y_i and
 * y_j (the form pcoa_compute takes from N = 16,384; needs N % 4 == 0). */
int pcoa_debug_centred_matvec(pcoa_ctx* ctx, const double* x, double* y, int upper_triangle_form);
int pcoa_debug_free(void* p);
int pcoa_debug_guard_mode(void);

/* Name of the GPU the ctx runs on, its CU count and the library version string. */
int pcoa_device_info(pcoa_ctx* ctx, char* name_out, int32_t name_cap, int32_t* cu_count_out);
const char* pcoa_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PCOA_H_ */
