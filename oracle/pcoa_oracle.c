/*
 * pcoa_oracle.c -- CPU restatement of the reference PCoA hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the *checker*, never the product: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The shipped path (spark-examples_amd/csrc) never
 * links or calls anything in oracle/.
 *
 * PARITY STATUS: the reference repository holds no tests, golden vectors or fixtures for this path
 * (SURVEY.md section 4), and its Scala/Spark build cannot be compiled here (no JVM).  The Gram and
 * centring stages are pinned against the reference's own Python twin (variants_pca.py), executed in
 * this container through tests/golden/make_golden.py (fixtures under tests/golden/).  The PCA stage
 * (Spark MLlib 1.6.1 RowMatrix.computePrincipalComponents, a third-party dependency absent from
 * /root/reference) is restated from its published algorithm => for that stage: "parity unpinned".
 *
 * Each function cites the reference lines it follows.  Paths are relative to /root/reference:
 *   VariantsPca.scala = src/main/scala/com/google/cloud/genomics/spark/examples/VariantsPca.scala
 *   variants_pca.py   = src/main/python/variants_pca.py
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/*
 * getSimilarityMatrix -- VariantsPca.scala:182-191 (twin: variants_pca.py:54-82).
 *
 * "Partition" p takes a contiguous range of variants, allocates a dense N x N Int matrix
 * (DenseMatrix.zeros[Int], :185) and for every variant and every ORDERED pair (c1, c2) of its
 * carriers does matrix(c1,c2) += 1 (:186-188).  Partials are summed (reduceByKey(_ + _), :190).
 * Scala Int arithmetic wraps at 2^31, reproduced here with uint32 adds.
 *
 * n_partitions plays the role of the RDD partition count; the result does not depend on it.
 * out: N x N row-major int32.
 */
int oracle_similarity_csr(const int32_t* sample_idx, const int64_t* row_offsets, int64_t n_variants,
                          int32_t n, int32_t n_partitions, int32_t* out) {
  if (n <= 0 || n_partitions <= 0) return -1;
  const size_t nn = (size_t)n * (size_t)n;
  uint32_t* partials = (uint32_t*)calloc(nn * (size_t)n_partitions, sizeof(uint32_t));
  if (!partials) return -2;
  int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
  for (int32_t p = 0; p < n_partitions; ++p) {
    uint32_t* m = partials + nn * (size_t)p;
    const int64_t v0 = n_variants * p / n_partitions;
    const int64_t v1 = n_variants * (p + 1) / n_partitions;
    for (int64_t v = v0; v < v1; ++v) {
      const int64_t b = row_offsets[v], e = row_offsets[v + 1];
      for (int64_t a1 = b; a1 < e; ++a1) {
        const int32_t c1 = sample_idx[a1];
        if (c1 < 0 || c1 >= n) { bad = 1; continue; }
        uint32_t* row = m + (size_t)c1 * n;
        for (int64_t a2 = b; a2 < e; ++a2) {
          const int32_t c2 = sample_idx[a2];
          if (c2 < 0 || c2 >= n) { bad = 1; continue; }
          row[c2] += 1u; /* matrix.update(c1, c2, matrix(c1, c2) + 1), :188 */
        }
      }
    }
  }
  /* reduceByKey(_ + _), :190 */
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < nn; ++i) {
    uint32_t s = 0;
    for (int32_t p = 0; p < n_partitions; ++p) s += partials[nn * (size_t)p + i];
    out[i] = (int32_t)s;
  }
  free(partials);
  return bad ? -3 : 0;
}

/*
 * Dense twin of the above for a variants x samples fp32 tile holding carrier multiplicities
 * (0/1 for well-formed input: x[v,i] = 1 iff extractCallInfo's hasVariation, VariantsPca.scala:56-60).
 * Rows with no carrier are dropped exactly as getCallsRdd does (:164-167) -- they add nothing.
 * Accumulates in int64 so it can also check runs longer than 2^31 variants.
 */
int oracle_similarity_dense_f32(const float* x, int64_t n_variants, int64_t ld, int32_t n,
                                int64_t* out) {
  if (n <= 0 || ld < n) return -1;
  const size_t nn = (size_t)n * (size_t)n;
  int nthreads = 1;
#ifdef _OPENMP
  nthreads = omp_get_max_threads();
#endif
  int64_t* partials = (int64_t*)calloc(nn * (size_t)nthreads, sizeof(int64_t));
  if (!partials) return -2;
#pragma omp parallel
  {
    int t = 0;
#ifdef _OPENMP
    t = omp_get_thread_num();
#endif
    int64_t* m = partials + nn * (size_t)t;
    int32_t* carriers = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
    int32_t* mult = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
#pragma omp for schedule(static)
    for (int64_t v = 0; v < n_variants; ++v) {
      const float* row = x + (size_t)v * (size_t)ld;
      int32_t k = 0;
      for (int32_t i = 0; i < n; ++i)
        if (row[i] != 0.0f) { carriers[k] = i; mult[k] = (int32_t)row[i]; ++k; }
      for (int32_t a = 0; a < k; ++a) {
        int64_t* mr = m + (size_t)carriers[a] * n;
        const int64_t ma = mult[a];
        for (int32_t b = 0; b < k; ++b) mr[carriers[b]] += ma * mult[b];
      }
    }
    free(carriers);
    free(mult);
  }
#pragma omp parallel for schedule(static)
  for (size_t i = 0; i < nn; ++i) {
    int64_t s = 0;
    for (int t = 0; t < nthreads; ++t) s += partials[nn * (size_t)t + i];
    out[i] = s;
  }
  free(partials);
  return 0;
}

/*
 * computePca part 1+2 -- VariantsPca.scala:199-223 (twin: variants_pca.py:84-121).
 *
 *   rowSums(i)  = entries(i).foldLeft(0D)(_ + _._2)          (:206)  left-to-right fp64 fold, j ascending
 *   nonZeroRows = rowSums.filter(_ > 0).size                 (:207)
 *   matrixSum   = rowSums.reduce(_ + _)                      (:210)  left-to-right
 *   matrixMean  = matrixSum / rowCount / rowCount            (:211)  two divisions, in that order
 *   rowMean     = rowSums(i) / rowCount                      (:216)
 *   colMean     = rowSums(j) / rowCount                      (:220)
 *   B(i,j)      = data - rowMean - colMean + matrixMean      (:221)  ((data-rowMean)-colMean)+matrixMean
 *
 * Compile with -ffp-contract=off: the JVM never fuses multiply-add.
 */
int oracle_center(const int64_t* s, int32_t n, double* b_out, double* row_sums_out,
                  int32_t* nonzero_rows_out, double* matrix_mean_out) {
  if (n <= 0) return -1;
  double* rs = (double*)malloc(sizeof(double) * (size_t)n);
  if (!rs) return -2;
  int32_t nz = 0;
  for (int32_t i = 0; i < n; ++i) {
    double acc = 0.0;
    for (int32_t j = 0; j < n; ++j) acc = acc + (double)s[(size_t)i * n + j];
    rs[i] = acc;
    if (acc > 0.0) ++nz;
  }
  double msum = rs[0];
  for (int32_t i = 1; i < n; ++i) msum = msum + rs[i];
  const double rc = (double)n;
  const double mmean = msum / rc / rc;
#pragma omp parallel for schedule(static)
  for (int32_t i = 0; i < n; ++i) {
    const double row_mean = rs[i] / rc;
    for (int32_t j = 0; j < n; ++j) {
      const double col_mean = rs[j] / rc;
      const double data = (double)s[(size_t)i * n + j];
      b_out[(size_t)i * n + j] = data - row_mean - col_mean + mmean;
    }
  }
  if (row_sums_out) memcpy(row_sums_out, rs, sizeof(double) * (size_t)n);
  if (nonzero_rows_out) *nonzero_rows_out = nz;
  if (matrix_mean_out) *matrix_mean_out = mmean;
  free(rs);
  return 0;
}

/*
 * computePca part 3, covariance half -- VariantsPca.scala:224-226 calls
 * org.apache.spark.mllib.linalg.distributed.RowMatrix.computePrincipalComponents (Spark MLlib 1.6.1,
 * build.sbt:11,25; not vendored under /root/reference).  Restated from the published 1.6.1 source:
 *
 *   computeCovariance():
 *     (m, mean) = treeAggregate: m = number of rows, mean = sum of rows;  mean :/= m
 *     G = computeGramianMatrix()   -- sum over rows r of r r^T (BLAS.spr into a packed upper triangle)
 *     m1 = m - 1.0
 *     for i: alpha = m / m1 * mean(i); for j >= i: G(i,j) = G(j,i) = G(i,j) / m1 - alpha * mean(j)
 *   computePrincipalComponents(k): SVD(Cov) via Breeze svd -> LAPACK dgesdd; first k columns of U.
 *
 * This function produces Cov (row-major N x N, symmetric); the SVD is done by the caller with
 * LAPACK dgesdd (numpy.linalg.svd), the same routine Breeze dispatches to.
 * The row-sum order inside G differs from Spark's partition-dependent treeAggregate order, which
 * the reference itself does not fix.
 */
int oracle_mllib_covariance(const double* b, int32_t n, double* cov_out) {
  if (n <= 1) return -1;
  double* mean = (double*)calloc((size_t)n, sizeof(double));
  if (!mean) return -2;
  for (int32_t r = 0; r < n; ++r)
    for (int32_t j = 0; j < n; ++j) mean[j] = mean[j] + b[(size_t)r * n + j];
  const double m = (double)n;
  for (int32_t j = 0; j < n; ++j) mean[j] = mean[j] / m;
  /* Gramian G = B^T B, upper triangle, accumulated row by row (spr: G += r r^T), i.e. every
   * G(i,j) is the r-ascending sum of b(r,i)*b(r,j).  B is transposed first only so that the
   * r-loop walks contiguous memory; the summation order is unchanged. */
  double* bt = (double*)malloc(sizeof(double) * (size_t)n * (size_t)n);
  if (!bt) { free(mean); return -2; }
  for (int32_t r = 0; r < n; ++r)
    for (int32_t j = 0; j < n; ++j) bt[(size_t)j * n + r] = b[(size_t)r * n + j];
#pragma omp parallel for schedule(dynamic, 8)
  for (int32_t i = 0; i < n; ++i) {
    const double* bi = bt + (size_t)i * n;
    for (int32_t j = i; j < n; ++j) {
      const double* bj = bt + (size_t)j * n;
      double acc = 0.0;
      for (int32_t r = 0; r < n; ++r) acc = acc + bi[r] * bj[r];
      cov_out[(size_t)i * n + j] = acc;
    }
  }
  free(bt);
  const double m1 = m - 1.0;
  for (int32_t i = 0; i < n; ++i) {
    const double alpha = m / m1 * mean[i];
    for (int32_t j = i; j < n; ++j) {
      const double gij = cov_out[(size_t)i * n + j] / m1 - alpha * mean[j];
      cov_out[(size_t)i * n + j] = gij;
      cov_out[(size_t)j * n + i] = gij;
    }
  }
  free(mean);
  return 0;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

void oracle_set_num_threads(int n) {
#ifdef _OPENMP
  omp_set_num_threads(n > 0 ? n : 1);
#else
  (void)n;
#endif
}
