"""CPU oracle for the PCoA hot path -- TEST INFRASTRUCTURE ONLY (see pcoa_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It is the checker, never the thing measured as the product or shipped.

PARITY STATUS
  * Gram + centring: pinned against the reference's own Python twin (variants_pca.py) run through
    tests/golden/make_golden.py; fixtures in tests/golden/*.npz.
  * PCA stage (Spark MLlib 1.6.1 RowMatrix.computePrincipalComponents -> Breeze svd -> LAPACK
    dgesdd; build.sbt:11,25, not under /root/reference): restated from the published algorithm,
    "parity unpinned" by any reference-side vector.

Function names mirror the reference twin /root/reference/src/main/python/variants_pca.py:
  calculate_similarity_matrix (:54-82), center_matrix (:84-121), perform_pca (:123-152)
and the Scala driver VariantsPca.scala:182-231.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile libpcoa_oracle.so (gcc, OpenMP).  Building the checker is not using it."""
    so = os.path.join(_HERE, "libpcoa_oracle.so")
    src = os.path.join(_HERE, "pcoa_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libpcoa_oracle.so"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        so = build()
        lib = ctypes.CDLL(so)
        i32p = ctypes.POINTER(ctypes.c_int32)
        i64p = ctypes.POINTER(ctypes.c_int64)
        f32p = ctypes.POINTER(ctypes.c_float)
        f64p = ctypes.POINTER(ctypes.c_double)
        lib.oracle_similarity_csr.argtypes = [i32p, i64p, ctypes.c_int64, ctypes.c_int32,
                                              ctypes.c_int32, i32p]
        lib.oracle_similarity_csr.restype = ctypes.c_int
        lib.oracle_similarity_dense_f32.argtypes = [f32p, ctypes.c_int64, ctypes.c_int64,
                                                    ctypes.c_int32, i64p]
        lib.oracle_similarity_dense_f32.restype = ctypes.c_int
        lib.oracle_center.argtypes = [i64p, ctypes.c_int32, f64p, f64p, i32p, f64p]
        lib.oracle_center.restype = ctypes.c_int
        lib.oracle_mllib_covariance.argtypes = [f64p, ctypes.c_int32, f64p]
        lib.oracle_mllib_covariance.restype = ctypes.c_int
        lib.oracle_num_threads.restype = ctypes.c_int
        lib.oracle_set_num_threads.argtypes = [ctypes.c_int]
        _LIB = lib
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def num_threads():
    return int(_lib().oracle_num_threads())


def set_num_threads(n):
    _lib().oracle_set_num_threads(int(n))


def callsets_to_csr(callsets):
    """List of per-variant carrier lists (RDD[Seq[Int]], VariantsPca.scala:153-168) -> CSR arrays."""
    offs = np.zeros(len(callsets) + 1, dtype=np.int64)
    for v, c in enumerate(callsets):
        offs[v + 1] = offs[v] + len(c)
    idx = np.zeros(max(int(offs[-1]), 1), dtype=np.int32)
    for v, c in enumerate(callsets):
        idx[offs[v]:offs[v + 1]] = np.asarray(c, dtype=np.int32)
    return idx, offs


def calculate_similarity_matrix(callsets, matrix_size, n_partitions=4):
    """S = sum over variants of the ordered-pair indicator (VariantsPca.scala:182-191;
    variants_pca.py:54-82).  Returns N x N int32 (Scala Int semantics, wraps at 2^31)."""
    if isinstance(callsets, tuple):
        idx, offs = callsets
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        offs = np.ascontiguousarray(offs, dtype=np.int64)
    else:
        idx, offs = callsets_to_csr(callsets)
    n = int(matrix_size)
    out = np.zeros((n, n), dtype=np.int32)
    rc = _lib().oracle_similarity_csr(_p(idx, ctypes.c_int32), _p(offs, ctypes.c_int64),
                                      len(offs) - 1, n, int(n_partitions), _p(out, ctypes.c_int32))
    if rc == -3:
        # the reference throws NoSuchElementException / ArrayIndexOutOfBounds on a bad index
        raise IndexError("callset index out of range")
    if rc != 0:
        raise RuntimeError("oracle_similarity_csr failed: %d" % rc)
    return out


def similarity_matrix_python_loops(callsets, matrix_size):
    """Literal transcription of the triple loop in variants_pca.py:67-72 (small cases only)."""
    m = np.zeros((matrix_size, matrix_size), dtype=np.int64)
    for callset in callsets:
        for x in callset:
            for y in callset:
                m[y][x] += 1
    return m


def similarity_from_dense(x, n_samples=None):
    """Dense variants x samples fp32 tile -> N x N int64 via the faithful pair loop."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    v, ld = x.shape
    n = int(n_samples) if n_samples is not None else ld
    out = np.zeros((n, n), dtype=np.int64)
    rc = _lib().oracle_similarity_dense_f32(_p(x, ctypes.c_float), v, ld, n, _p(out, ctypes.c_int64))
    if rc != 0:
        raise RuntimeError("oracle_similarity_dense_f32 failed: %d" % rc)
    return out


def similarity_from_dense_blas(x):
    """Best-effort CPU baseline: sgemm X^T X in chunks that keep fp32 counts exact (< 2^24),
    folded into int64.  Not the reference's algorithm; used only as the 'honest CPU number'."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    v, n = x.shape
    out = np.zeros((n, n), dtype=np.int64)
    step = 1 << 20
    for v0 in range(0, v, step):
        xs = x[v0:v0 + step]
        out += np.rint(xs.T @ xs).astype(np.int64)
    return out


def center_matrix(sim_matrix):
    """Row sums, matrix mean and double-centring in the reference's evaluation order
    (VariantsPca.scala:199-223; variants_pca.py:84-121).
    Returns (B, row_sums, nonzero_rows, matrix_mean)."""
    s = np.ascontiguousarray(sim_matrix, dtype=np.int64)
    n = s.shape[0]
    b = np.zeros((n, n), dtype=np.float64)
    rs = np.zeros(n, dtype=np.float64)
    nz = ctypes.c_int32(0)
    mm = ctypes.c_double(0.0)
    rc = _lib().oracle_center(_p(s, ctypes.c_int64), n, _p(b, ctypes.c_double),
                              _p(rs, ctypes.c_double), ctypes.byref(nz), ctypes.byref(mm))
    if rc != 0:
        raise RuntimeError("oracle_center failed: %d" % rc)
    return b, rs, int(nz.value), float(mm.value)


def mllib_covariance(b):
    """RowMatrix.computeCovariance of Spark MLlib 1.6.1 (see pcoa_oracle.c)."""
    b = np.ascontiguousarray(b, dtype=np.float64)
    n = b.shape[0]
    cov = np.zeros((n, n), dtype=np.float64)
    rc = _lib().oracle_mllib_covariance(_p(b, ctypes.c_double), n, _p(cov, ctypes.c_double))
    if rc != 0:
        raise RuntimeError("oracle_mllib_covariance failed: %d" % rc)
    return cov


def perform_pca(b, nr_principal_components=2):
    """RowMatrix.computePrincipalComponents(k) (called at VariantsPca.scala:224-226 and
    variants_pca.py:147-150): Cov -> LAPACK dgesdd -> first k columns of U.
    Returns (components N x k, singular_values k)."""
    n = b.shape[0]
    k = int(nr_principal_components)
    if not (0 < k <= n):
        # MLlib: require(k > 0 && k <= n)
        raise ValueError("k = %d out of range (0, n = %d]" % (k, n))
    cov = mllib_covariance(b)
    u, s, _ = np.linalg.svd(cov)  # numpy -> LAPACK *gesdd, as Breeze svd
    return np.ascontiguousarray(u[:, :k]), s[:k].copy()


def sign_normalize(vecs):
    """Flip each column so that its entry of largest magnitude is positive (ties -> lowest index).
    The reference's sign is whatever LAPACK returns; parity is compared after this."""
    v = np.array(vecs, dtype=np.float64, copy=True)
    if v.ndim == 1:
        v = v[:, None]
    for c in range(v.shape[1]):
        i = int(np.argmax(np.abs(v[:, c])))
        if v[i, c] < 0:
            v[:, c] = -v[:, c]
    return v


def compute_pca(sim_matrix, num_pc=2):
    """computePca end to end (VariantsPca.scala:198-231): returns dict with components (N x k,
    sign-normalised), eigenvalues of B implied by the SVD of Cov (lambda = sqrt(s * (N-1))),
    nonzero_rows, B."""
    b, rs, nz, mm = center_matrix(sim_matrix)
    comps, svals = perform_pca(b, num_pc)
    n = b.shape[0]
    lam = np.sqrt(np.maximum(svals, 0.0) * (n - 1.0))
    return {"components": sign_normalize(comps), "eigenvalues": lam, "nonzero_rows": nz,
            "row_sums": rs, "matrix_mean": mm, "B": b}


def emit_result(result_rows, names=None):
    """emitResult stdout format (VariantsPca.scala:233-239): name \\t dataset \\t pc1 \\t pc2,
    sorted by name; dataset = callsetId.split('-').head."""
    rows = []
    for callset_id, pc1, pc2 in result_rows:
        dataset = callset_id.split("-")[0]
        name = names[callset_id] if names else callset_id
        rows.append((name, pc1, pc2, dataset))
    rows.sort(key=lambda t: t[0])
    return ["%s\t%s\t%s\t%s" % (r[0], r[3], repr(float(r[1])), repr(float(r[2]))) for r in rows]
