"""numpy stand-ins for the GPU strip owners (TEST INFRASTRUCTURE: the CPU tests of strips.py and the --standin mode of
tools/config5_strips.py).  Same interface as PcoaEngine(strip=(col0, cols)): .n, .strip, .accumulate_bits, .strip_col_sums,
.strip_matvec -- and .lanczos, the eigensolver the product gets from the C ABI (pcoa_lanczos_with_matvec: the engine's Krylov
iteration on the GPU), restated here in numpy so that the control flow of compute_pca_over_strips (gathers, feeding, ragged
strips over gloo ranks) can run without a GPU."""
import numpy as np


def _sign_normalize(u):
    """largest-magnitude entry positive, ties -> lowest index (the engine's and the oracle's convention)"""
    u = np.array(u, dtype=np.float64, copy=True)
    for c in range(u.shape[1]):
        i = int(np.argmax(np.abs(u[:, c])))
        if u[i, c] < 0:
            u[:, c] = -u[:, c]
    return u



class HostStrip(object):
    """holds S[:, col0:col0+cols] and evaluates the rows of B in the reference's operation order, as csrc/center.hip:
    strip_band_kernel does."""

    def __init__(self, s_full, col0, cols):
        self.n = s_full.shape[0]
        self.strip = (col0, cols)
        self.s = np.ascontiguousarray(s_full[:, col0:col0 + cols]).astype(np.float64)

    @classmethod
    def empty(cls, n, col0, cols):
        return cls(np.zeros((n, n), dtype=np.int64), col0, cols)

    def accumulate_bits(self, bits):
        """carrier bitsets [v][ceil(n/32)] uint32 (pcoa_accumulate_bits): S[:, strip] += X^T X[:, strip]"""
        bits = np.asarray(bits)
        assert bits.dtype == np.uint32 and bits.ndim == 2 and bits.shape[1] == (self.n + 31) // 32
        x = ((bits[:, :, None] >> np.arange(32, dtype=np.uint32)[None, None, :]) & 1).reshape(bits.shape[0], -1)[:, :self.n]
        x = x.astype(np.float64)
        col0, cols = self.strip
        self.s += x.T @ x[:, col0:col0 + cols]

    def strip_col_sums(self):
        return self.s.sum(axis=0)

    def strip_matvec(self, v, means, matrix_mean):
        col0, cols = self.strip
        b = ((self.s - means[col0:col0 + cols][None, :]) - means[:, None]) + matrix_mean   # B(j, i) at [i, jj]
        return b.T @ v

    def close(self):
        pass

    def lanczos(self, matvec, num_pc, max_steps=512, tol=1e-11, first_check=12, trace=None):
        """Top-k eigenpairs of the operator `matvec` (numpy N-vector -> numpy N-vector): Lanczos with full re-orthogonalisation,
        Ritz pairs accepted on their TRUE residual -- the algorithm of csrc/eig_lanczos.hip in numpy.  Returns (components
        [N][k] sign-normalised unit columns, eigenvalues [k]); RuntimeError when no verified pair is reached."""
        n, k = int(self.n), int(num_pc)
        # deterministic start vector (the same LCG stream on every rank)
        idx = np.arange(1, n + 1, dtype=np.uint64)
        s = (idx * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
        s = (s * np.uint64(1103515245) + np.uint64(12345)) & np.uint64(0xFFFFFFFF)
        s ^= s >> np.uint64(15)
        s = (s * np.uint64(1103515245) + np.uint64(12345)) & np.uint64(0xFFFFFFFF)
        v = (s >> np.uint64(8)).astype(np.float64) * (1.0 / 8388608.0) - 1.0
        v /= np.linalg.norm(v)

        mmax = int(min(max_steps, n))
        basis = np.zeros((mmax + 1, n), dtype=np.float64)
        basis[0] = v
        norm = lambda t: float(np.linalg.norm(t))                      # noqa: E731
        alpha, beta = [], []
        next_check = max(k + 1, min(first_check, mmax))
        for j in range(mmax):
            w = matvec(basis[j])
            a = float(basis[j] @ w)
            alpha.append(a)
            # full re-orthogonalisation, twice (classical Gram-Schmidt x 2)
            for _ in range(2):
                w -= basis[:j + 1].T @ (basis[:j + 1] @ w)
            b = norm(w)
            m = j + 1
            breakdown = b <= 1e-14 * max(1.0, max(abs(x) for x in alpha))
            if m >= next_check or breakdown or m == mmax:
                t = np.diag(alpha) + np.diag(beta, 1) + np.diag(beta, -1)
                lam, y = np.linalg.eigh(t)
                order = np.argsort(-lam)[:k]
                theta = lam[order]
                scale = float(np.max(np.abs(lam)))
                est = np.abs(b * y[-1, order])
                gaps = np.array([np.min(np.abs(np.delete(lam, order[c]) - theta[c])) if m > 1 else scale for c in range(len(order))])
                ok = len(order) == k and bool(np.all(est <= tol * scale) and np.all(est <= 1e-8 * gaps))
                if trace is not None:
                    trace.append((m, theta.copy(), est.copy()))
                if ok or breakdown:
                    yk = np.ascontiguousarray(y[:, order])
                    u = basis[:m].T @ yk
                    u = u / np.linalg.norm(u, axis=0)
                    # accept only on the TRUE residual (one more mat-vec per vector)
                    good = len(order) == k
                    for c in range(len(order)):
                        uc = u[:, c]
                        res = norm(matvec(uc) - theta[c] * uc)
                        good = good and res <= max(tol * scale * 10.0, 1e-9 * abs(theta[c])) and res <= 1e-6 * max(gaps[c], 1e-300)
                    if good:
                        return _sign_normalize(u), theta.copy()
                next_check = m + (4 if m < 24 else 8)
            if breakdown or m == mmax:
                break
            beta.append(b)
            basis[j + 1] = w / b
        raise RuntimeError("Lanczos over strips did not reach a verified residual in %d steps (tiny spectral gaps?); "
                           "there is no dense fallback for a matrix tiled across GPUs" % len(alpha))
