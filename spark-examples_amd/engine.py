"""PcoaEngine -- thin object wrapper over the C ABI (include/pcoa.h).

One engine = one pcoa_ctx = one GPU.  The method names follow the stages of the reference driver
(VariantsPca.scala:44-47): getSimilarityMatrix -> accumulate_* / finalize, computePca -> compute.
"""
import ctypes

import numpy as np

from . import _lib as L


class PcoaError(RuntimeError):
    def __init__(self, code, message):
        RuntimeError.__init__(self, "pcoa error %d: %s" % (code, message))
        self.code = code


class IndexRangeError(PcoaError, IndexError):
    """A callset index outside [0, N): the reference throws NoSuchElementException /
    IndexOutOfBounds here (VariantsPca.scala:59, :188)."""


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


class PcoaEngine(object):
    def __init__(self, n_samples, device=0, flags=L.PCOA_FLAG_DEFAULT, gram_kernel=None, eig=None, strip=None,
                 pipeline=True, operand=None):
        """gram_kernel: None/"auto" (MX-FP4 MFMA for binary tiles, int8 MFMA for multiplicities; both exact),
        "fp4", "i8" (force one of them) or "f32" (fp32-MFMA path).
        eig: None/"auto" (Lanczos with verified residual -- single vector, then the band iteration --, Householder fallback),
        "householder", "lanczos" (no dense fallback), "band" (Lanczos as the band iteration from the start, PCOA_FLAG_EIG_BAND).
        strip: None, or (col0, cols): a strip owner holding S[:, col0:col0+cols] (pcoa_create_strip; see strips.py).
        pipeline=False: PCOA_FLAG_NO_PIPELINE (fp32 pre-pass and contraction strictly serial; measurements).
        operand: None/"bits" (binary tiles are re-laid out to 1 bit per genotype and expanded to MX-FP4 inside the
        contraction) or "fp4" (PCOA_FLAG_OPERAND_FP4: the operand is stored as MX-FP4, 4 bits per genotype)."""
        if not pipeline:
            flags |= L.PCOA_FLAG_NO_PIPELINE
        if operand == "fp4":
            flags |= L.PCOA_FLAG_OPERAND_FP4
        elif operand not in (None, "bits"):
            raise ValueError("operand must be 'bits' or 'fp4'")
        if eig == "householder":
            flags |= L.PCOA_FLAG_EIG_HOUSEHOLDER
        elif eig == "lanczos":
            flags |= L.PCOA_FLAG_EIG_LANCZOS
        elif eig == "band":   # Lanczos only, as the band iteration from the start (eigenvalues of exact multiplicity > 1)
            flags |= L.PCOA_FLAG_EIG_LANCZOS | L.PCOA_FLAG_EIG_BAND
        elif eig not in (None, "auto"):
            raise ValueError("eig must be 'auto', 'householder', 'lanczos' or 'band'")
        if gram_kernel == "f32":
            flags |= L.PCOA_FLAG_GRAM_F32_MFMA
        elif gram_kernel == "i8":
            flags |= L.PCOA_FLAG_GRAM_I8_MFMA
        elif gram_kernel == "fp4":
            flags |= L.PCOA_FLAG_GRAM_FP4_MFMA
        elif gram_kernel not in (None, "auto"):
            raise ValueError("gram_kernel must be 'auto', 'fp4', 'i8' or 'f32'")
        self._lib = L.load()
        self._ctx = ctypes.c_void_p()
        self.strip = None if strip is None else (int(strip[0]), int(strip[1]))
        if self.strip is None:
            rc = self._lib.pcoa_create(ctypes.byref(self._ctx), int(n_samples), int(device), int(flags))
        else:
            rc = self._lib.pcoa_create_strip(ctypes.byref(self._ctx), int(n_samples), self.strip[0], self.strip[1],
                                             int(device), int(flags))
        if rc != L.PCOA_OK:
            msg = self._lib.pcoa_last_error(None)
            self._ctx = None
            raise PcoaError(rc, msg.decode() if msg else "pcoa_create failed")
        self.n = int(n_samples)
        self.cols = self.n if self.strip is None else self.strip[1]   # columns of S this ctx holds
        self.device = int(device)
        self._keepalive = []

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc):
        if rc != L.PCOA_OK:
            msg = self._lib.pcoa_last_error(self._ctx)
            msg = msg.decode() if msg else ""
            if rc == L.PCOA_ERR_INDEX_RANGE:
                raise IndexRangeError(rc, msg)
            raise PcoaError(rc, msg)

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.pcoa_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def sync(self):
        self._check(self._lib.pcoa_sync(self._ctx))
        self._keepalive = []

    def reset(self):
        self._check(self._lib.pcoa_reset(self._ctx))

    def reserve(self, variants_per_call=0, num_pc=0):
        """Allocate now what the first accumulate / compute calls would allocate lazily (pcoa_reserve)."""
        self._check(self._lib.pcoa_reserve(self._ctx, int(variants_per_call), int(num_pc)))

    def set_stream(self, hip_stream_ptr):
        self._check(self._lib.pcoa_set_stream(self._ctx, ctypes.c_void_p(hip_stream_ptr or 0)))

    def device_info(self):
        buf = ctypes.create_string_buffer(256)
        cu = ctypes.c_int32(0)
        self._check(self._lib.pcoa_device_info(self._ctx, buf, 256, ctypes.byref(cu)))
        return buf.value.decode(), int(cu.value)

    # ------------------------------------------------------------------ getSimilarityMatrix
    def accumulate_calls(self, sample_idx, row_offsets):
        """CSR carrier lists: exactly what RDD[Seq[Int]] carries (VariantsPca.scala:153-168)."""
        idx = np.ascontiguousarray(sample_idx, dtype=np.int32)
        offs = np.ascontiguousarray(row_offsets, dtype=np.int64)
        if offs.ndim != 1 or offs.size < 1:
            raise ValueError("row_offsets must have n_variants + 1 entries")
        # the C side walks sample_idx[row_offsets[0] .. row_offsets[n]) through raw pointers: inconsistent arrays
        # must fail here, not as an out-of-bounds read
        if idx.ndim != 1:
            raise ValueError("sample_idx must be one-dimensional")
        if offs[0] < 0 or offs[-1] > idx.size or (offs.size > 1 and np.any(np.diff(offs) < 0)):
            raise ValueError("row_offsets must be non-decreasing, start at >= 0 and end at <= len(sample_idx) = %d"
                             % idx.size)
        if idx.size == 0:
            idx = np.zeros(1, dtype=np.int32)
        self._check(self._lib.pcoa_accumulate_calls(self._ctx, _ptr(idx), _ptr(offs), offs.size - 1))

    def accumulate_calls_tensors(self, sample_idx, row_offsets, asynchronous=False):
        """The same boundary for torch tensors (plumbing: memory handles only) -- pcoa_accumulate_calls_ex:
        CUDA tensors are read in place (PCOA_CALLS_DEVICE_PTR), pinned CPU tensors by the DMA engine (PCOA_CALLS_HOST_PINNED),
        other CPU tensors like numpy arrays.  asynchronous=True (PCOA_CALLS_ASYNC): return when the work is queued; the tensors
        are kept alive (and must stay unmodified) until the next synchronising call."""
        import torch
        assert sample_idx.dtype == torch.int32 and row_offsets.dtype == torch.int64
        assert sample_idx.dim() == 1 and row_offsets.dim() == 1 and sample_idx.is_contiguous() and row_offsets.is_contiguous()
        assert sample_idx.device == row_offsets.device
        if sample_idx.numel() == 0:
            return   # only empty lists: they add nothing (filtered at VariantsPca.scala:166); an empty tensor has no address
        flags = 0
        if sample_idx.is_cuda:
            flags |= L.PCOA_CALLS_DEVICE_PTR
            torch.cuda.current_stream(sample_idx.device).synchronize()   # the engine runs on its own stream
        elif sample_idx.is_pinned() and row_offsets.is_pinned():
            flags |= L.PCOA_CALLS_HOST_PINNED
        if asynchronous and not sample_idx.is_cuda:
            flags |= L.PCOA_CALLS_ASYNC
        if flags & (L.PCOA_CALLS_DEVICE_PTR | L.PCOA_CALLS_ASYNC):
            self._keepalive.append((sample_idx, row_offsets))
        self._check(self._lib.pcoa_accumulate_calls_ex(self._ctx, ctypes.c_void_p(sample_idx.data_ptr()),
                                                       ctypes.c_void_p(row_offsets.data_ptr()), int(row_offsets.shape[0]) - 1, flags))

    def accumulate_callsets(self, callsets):
        """Convenience: list of per-variant index lists."""
        offs = np.zeros(len(callsets) + 1, dtype=np.int64)
        for v, c in enumerate(callsets):
            offs[v + 1] = offs[v] + len(c)
        idx = np.fromiter((i for c in callsets for i in c), dtype=np.int32, count=int(offs[-1]))
        self.accumulate_calls(idx, offs)

    def accumulate_dense(self, x, n_variants=None, ld=None):
        """Dense variants x samples fp32 tile.  `x` is a numpy array (host) or any object with
        data_ptr()/is_cuda (a torch CUDA tensor: read in place, kept alive until sync())."""
        if hasattr(x, "data_ptr") and getattr(x, "is_cuda", False):
            import torch  # plumbing only: device memory handle
            assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
            nv = int(x.shape[0]) if n_variants is None else int(n_variants)
            ldv = int(x.stride(0)) if ld is None else int(ld)
            self._keepalive.append(x)
            # the engine runs on its own stream: whatever torch queued to produce x must have finished
            torch.cuda.current_stream(x.device).synchronize()
            self._check(self._lib.pcoa_accumulate_dense_f32(self._ctx, ctypes.c_void_p(x.data_ptr()), nv, ldv, 1))
            return
        a = np.ascontiguousarray(x, dtype=np.float32)
        if a.ndim != 2:
            raise ValueError("x must be 2-D [variants][samples]")
        nv = a.shape[0] if n_variants is None else int(n_variants)
        ldv = a.shape[1] if ld is None else int(ld)
        self._check(self._lib.pcoa_accumulate_dense_f32(self._ctx, _ptr(a), nv, ldv, 0))

    def accumulate_dense_u8(self, x, n_variants=None, ld=None):
        """Dense variants x samples tile with one byte per genotype (0..127): numpy uint8 array (host) or a
        torch uint8 CUDA tensor (read in place)."""
        if hasattr(x, "data_ptr") and getattr(x, "is_cuda", False):
            import torch  # plumbing only
            assert x.dtype == torch.uint8 and x.dim() == 2 and x.stride(1) == 1
            nv = int(x.shape[0]) if n_variants is None else int(n_variants)
            ldv = int(x.stride(0)) if ld is None else int(ld)
            self._keepalive.append(x)
            torch.cuda.current_stream(x.device).synchronize()  # see accumulate_dense
            self._check(self._lib.pcoa_accumulate_dense_u8(self._ctx, ctypes.c_void_p(x.data_ptr()), nv, ldv, 1))
            return
        a = np.ascontiguousarray(x, dtype=np.uint8)
        if a.ndim != 2:
            raise ValueError("x must be 2-D [variants][samples]")
        nv = a.shape[0] if n_variants is None else int(n_variants)
        ldv = a.shape[1] if ld is None else int(ld)
        self._check(self._lib.pcoa_accumulate_dense_u8(self._ctx, _ptr(a), nv, ldv, 0))

    def accumulate_bits(self, bits, n_variants=None, ld_words=None):
        """Bit-packed tile: row v = carrier bitset of variant v, sample i = bit (i & 31) of word i >> 5
        (`ingest.pack_bits` builds it).  numpy uint32 [V][W] (host) or a torch int32 CUDA tensor (read in place)."""
        if hasattr(bits, "data_ptr") and getattr(bits, "is_cuda", False):
            import torch  # plumbing only
            assert bits.dtype == torch.int32 and bits.dim() == 2 and bits.stride(1) == 1
            nv = int(bits.shape[0]) if n_variants is None else int(n_variants)
            ldv = int(bits.stride(0)) if ld_words is None else int(ld_words)
            self._keepalive.append(bits)
            torch.cuda.current_stream(bits.device).synchronize()  # see accumulate_dense
            self._check(self._lib.pcoa_accumulate_bits(self._ctx, ctypes.c_void_p(bits.data_ptr()), nv, ldv, 1))
            return
        a = np.ascontiguousarray(bits, dtype=np.uint32)
        if a.ndim != 2:
            raise ValueError("bits must be 2-D [variants][words]")
        nv = a.shape[0] if n_variants is None else int(n_variants)
        ldv = a.shape[1] if ld_words is None else int(ld_words)
        self._check(self._lib.pcoa_accumulate_bits(self._ctx, _ptr(a), nv, ldv, 0))

    def accumulate_plink_bed(self, rows, ref_is_a1=False, asynchronous=False):
        """Raw variant-major PLINK .bed rows, numpy uint8 [V][row_bytes] (host) or a torch uint8 CUDA tensor: decoded on the
        device (pcoa_accumulate_plink_bed).  asynchronous=True: a PINNED torch uint8 CPU tensor that is only queued
        (PCOA_BED_HOST_ASYNC): it must stay unmodified until the second later call has returned, or the next sync()."""
        if asynchronous:
            import torch  # plumbing only
            assert rows.dtype == torch.uint8 and rows.dim() == 2 and rows.stride(1) == 1 and rows.is_pinned()
            self._check(self._lib.pcoa_accumulate_plink_bed(self._ctx, ctypes.c_void_p(rows.data_ptr()), int(rows.shape[0]),
                                                            int(rows.stride(0)), int(bool(ref_is_a1)), L.PCOA_BED_HOST_ASYNC))
            return
        if hasattr(rows, "data_ptr") and getattr(rows, "is_cuda", False):
            import torch  # plumbing only
            assert rows.dtype == torch.uint8 and rows.dim() == 2 and rows.stride(1) == 1
            self._keepalive.append(rows)
            torch.cuda.current_stream(rows.device).synchronize()
            self._check(self._lib.pcoa_accumulate_plink_bed(self._ctx, ctypes.c_void_p(rows.data_ptr()), int(rows.shape[0]),
                                                            int(rows.stride(0)), int(bool(ref_is_a1)), 1))
            return
        a = np.ascontiguousarray(rows, dtype=np.uint8)
        if a.ndim != 2:
            raise ValueError("rows must be [variants][row_bytes]")
        self._check(self._lib.pcoa_accumulate_plink_bed(self._ctx, _ptr(a), a.shape[0], a.shape[1], int(bool(ref_is_a1)), 0))

    def accumulate_dense_device_ptr(self, ptr, n_variants, ld):
        self._check(self._lib.pcoa_accumulate_dense_f32(self._ctx, ctypes.c_void_p(int(ptr)), int(n_variants),
                                                        int(ld), 1))

    def _synth_params(self, seed, pop_offsets, thresholds):
        po = np.ascontiguousarray(pop_offsets, dtype=np.int32)
        th = np.ascontiguousarray(thresholds, dtype=np.uint32)
        p = L.PcoaSynthParams()
        p.seed = int(seed)
        p.n_pops = int(po.size - 1)
        p.pop_offsets = po.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))
        p.thresholds = th.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32))
        return p, (po, th)

    def accumulate_synthetic(self, seed, pop_offsets, thresholds, first_variant):
        th = np.asarray(thresholds)
        p, keep = self._synth_params(seed, pop_offsets, th)
        self._check(self._lib.pcoa_accumulate_synthetic(self._ctx, ctypes.byref(p), int(first_variant),
                                                        int(th.shape[0])))
        del keep

    def synth_fill(self, seed, pop_offsets, thresholds, first_variant, x_dev_ptr, ld):
        th = np.asarray(thresholds)
        p, keep = self._synth_params(seed, pop_offsets, th)
        self._check(self._lib.pcoa_synth_fill_f32(self._ctx, ctypes.byref(p), int(first_variant), int(th.shape[0]),
                                                  ctypes.c_void_p(int(x_dev_ptr)), int(ld)))
        del keep

    def finalize(self):
        self._check(self._lib.pcoa_gram_finalize(self._ctx))

    def gram(self):
        """All N^2 entries of S as int64 (zeros included, as matrix.iterator emits them, :189)."""
        out = np.zeros((self.n, self.cols), dtype=np.int64)   # a strip owner holds N x cols
        self._check(self._lib.pcoa_gram_read_i64(self._ctx, _ptr(out)))
        return out

    def gram_block(self, row0, col0, rows, cols):
        """The block S[row0:row0+rows, col0:col0+cols] as int64 (no N x N staging)."""
        out = np.zeros((int(rows), int(cols)), dtype=np.int64)
        self._check(self._lib.pcoa_gram_read_block_i64(self._ctx, int(row0), int(col0), int(rows), int(cols), _ptr(out)))
        return out

    def load_gram(self, s):
        a = np.ascontiguousarray(s, dtype=np.int64)
        if a.shape != (self.n, self.cols):
            raise ValueError("expected an %d x %d matrix" % (self.n, self.cols))
        self._check(self._lib.pcoa_gram_load_i64(self._ctx, _ptr(a)))

    def lanczos(self, matvec, num_pc, max_steps=512, tol=1e-11, first_check=12, trace=None):
        """Top-k eigenpairs of the symmetric operator `matvec` (float64 CUDA tensor of N entries on this engine's GPU -> the
        same) by the ENGINE's Lanczos iteration (pcoa_lanczos_with_matvec): Krylov basis, re-orthogonalisation, the Ritz
        problem and the true-residual acceptance run in the library's kernels; this method only wraps the callback's device
        pointers as tensors.  Returns (components [N][k] sign-normalised unit columns, eigenvalues [k]).  (max_steps / tol /
        first_check are the stand-in's knobs; the engine uses its own: 512, 1e-11, 12.)"""
        import torch  # plumbing only: a tensor view of a device pointer
        n, k = self.n, int(num_pc)
        dev = torch.device("cuda", self.device)

        class _Ptr(object):   # __cuda_array_interface__: N float64 at a raw device address
            def __init__(self, addr):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(addr), False), "version": 2}

        failure = []

        def call(_user, v_ptr, y_ptr):
            try:
                v = torch.as_tensor(_Ptr(v_ptr), device=dev)
                y = torch.as_tensor(_Ptr(y_ptr), device=dev)
                y.copy_(matvec(v))
                torch.cuda.current_stream(dev).synchronize()   # the engine reads y on its own stream next
                return 0
            except BaseException as exc:   # never unwind through the C frames
                failure.append(exc)
                return 1

        cb = L.MATVEC_FN(call)
        comps = np.zeros((k, n), dtype=np.float64)
        lam = np.zeros(k, dtype=np.float64)
        steps = ctypes.c_int32(0)
        rc = self._lib.pcoa_lanczos_with_matvec(self._ctx, k, cb, None, _ptr(comps), _ptr(lam), ctypes.byref(steps))
        if failure:
            raise failure[0]
        if trace is not None:
            trace.append((int(steps.value), lam.copy(), None))
        if rc == L.PCOA_ERR_NOT_CONVERGED:
            raise RuntimeError("Lanczos over strips did not reach a verified residual in %d steps (tiny spectral gaps?); there "
                               "is no dense fallback for a matrix tiled across GPUs" % int(steps.value))
        self._check(rc)
        return np.ascontiguousarray(comps.T), lam

    def debug_centred_matvec(self, x, upper_triangle_form):
        """One y = B x of the centred matrix of the current S (test hook of the two mat-vec forms of the eigensolver)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros(self.n, dtype=np.float64)
        self._check(self._lib.pcoa_debug_centred_matvec(self._ctx, _ptr(x), _ptr(y), int(upper_triangle_form)))
        return y

    def reduce_from(self, other):
        """self.S += other.S (two engines of this process; pcoa_gram_reduce_from: peer copy + int64 add, no collective)."""
        self._check(self._lib.pcoa_gram_reduce_from(self._ctx, other._ctx))

    def export_device(self, dst_ptr):
        self._check(self._lib.pcoa_gram_export_device_i64(self._ctx, ctypes.c_void_p(int(dst_ptr))))

    def import_device(self, src_ptr):
        self._check(self._lib.pcoa_gram_import_device_i64(self._ctx, ctypes.c_void_p(int(src_ptr))))

    # ------------------------------------------------------------------ native RCCL
    def comm_unique_id(self):
        buf = (ctypes.c_uint8 * 128)()
        self._check(self._lib.pcoa_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p)))
        return bytes(buf)

    def comm_init(self, unique_id, rank, n_ranks):
        buf = (ctypes.c_uint8 * 128).from_buffer_copy(unique_id)
        comm = ctypes.c_void_p()
        self._check(self._lib.pcoa_comm_init(self._ctx, ctypes.cast(buf, ctypes.c_void_p), int(rank), int(n_ranks),
                                             ctypes.byref(comm)))
        return comm

    def comm_runtime(self):
        """(path, version code) of the RCCL image the library's communicator calls are bound to."""
        buf = ctypes.create_string_buffer(1024)
        ver = ctypes.c_int32(0)
        self._check(self._lib.pcoa_comm_runtime(buf, 1024, ctypes.byref(ver)))
        return buf.value.decode(), int(ver.value)

    def comm_destroy(self, comm):
        self._check(self._lib.pcoa_comm_destroy(comm))

    def comm_count(self, comm):
        n = ctypes.c_int32(0)
        self._check(self._lib.pcoa_comm_count(comm, ctypes.byref(n)))
        return int(n.value)

    def allreduce_rccl(self, comm):
        self._check(self._lib.pcoa_gram_allreduce_rccl(self._ctx, comm))

    # ------------------------------------------------------------------ strip owner
    def strip_col_sums(self):
        """rowSums (VariantsPca.scala:206) of the strip's samples: column sums of the held N x cols block."""
        out = np.zeros(self.cols, dtype=np.float64)
        self._check(self._lib.pcoa_strip_col_sums(self._ctx, _ptr(out)))
        return out

    def strip_matvec(self, v, means, matrix_mean):
        """(B v)[col0:col0+cols] with B evaluated on the fly from the strip (VariantsPca.scala:216-221 order)."""
        v = np.ascontiguousarray(v, dtype=np.float64)
        means = np.ascontiguousarray(means, dtype=np.float64)
        if v.shape != (self.n,) or means.shape != (self.n,):
            raise ValueError("v and means must have N = %d entries" % self.n)
        out = np.zeros(self.cols, dtype=np.float64)
        self._check(self._lib.pcoa_strip_matvec(self._ctx, _ptr(v), _ptr(means), float(matrix_mean), _ptr(out)))
        return out

    def strip_set_centering(self, means, matrix_mean):
        """Row means and matrix mean of the current S, resident on the device for strip_matvec_device."""
        means = np.ascontiguousarray(means, dtype=np.float64)
        if means.shape != (self.n,):
            raise ValueError("means must have N = %d entries" % self.n)
        self._check(self._lib.pcoa_strip_set_centering(self._ctx, _ptr(means), float(matrix_mean)))

    def strip_matvec_device(self, v_dev):
        """(B v)[col0:col0+cols] for a float64 torch tensor v on this engine's GPU; returns a device tensor (no PCIe traffic)."""
        import torch  # plumbing only: device memory handles
        assert v_dev.is_cuda and v_dev.dtype == torch.float64 and v_dev.dim() == 1 and v_dev.shape[0] == self.n
        if v_dev.device.index != self.device:   # the ctx dereferences the pointer on ITS device
            raise ValueError("v lives on cuda:%d, this strip owner on cuda:%d" % (v_dev.device.index, self.device))
        v_dev = v_dev.contiguous()
        y = torch.empty(self.cols, dtype=torch.float64, device=v_dev.device)
        torch.cuda.current_stream(v_dev.device).synchronize()   # the engine runs on its own stream
        self._check(self._lib.pcoa_strip_matvec_device(self._ctx, ctypes.c_void_p(v_dev.data_ptr()), ctypes.c_void_p(y.data_ptr())))
        return y

    # ------------------------------------------------------------------ computePca
    def center(self, want_matrix=True):
        """Row sums + double-centring (VariantsPca.scala:206-223).  Returns (B, row_sums, nonzero_rows, mean)."""
        b = np.zeros((self.n, self.n), dtype=np.float64) if want_matrix else None
        rs = np.zeros(self.n, dtype=np.float64)
        nz = ctypes.c_int32(0)
        mm = ctypes.c_double(0.0)
        self._check(self._lib.pcoa_center_read_f64(self._ctx, _ptr(b) if want_matrix else None, _ptr(rs),
                                                   ctypes.byref(nz), ctypes.byref(mm)))
        return b, rs, int(nz.value), float(mm.value)

    def compute(self, num_pc=2):
        """computePca (VariantsPca.scala:198-231).  Returns (components [N, k], eigenvalues [k], nonzero_rows)."""
        k = int(num_pc)
        comps = np.zeros((max(k, 1), self.n), dtype=np.float64)  # column-major N x k == row-major k x N
        lam = np.zeros(max(k, 1), dtype=np.float64)
        nz = ctypes.c_int32(0)
        self._check(self._lib.pcoa_compute(self._ctx, k, _ptr(comps), _ptr(lam), ctypes.byref(nz)))
        return np.ascontiguousarray(comps[:k].T), lam[:k].copy(), int(nz.value)

    # ------------------------------------------------------------------ instrumentation
    def timings(self):
        t = L.PcoaTimings()
        self._check(self._lib.pcoa_get_timings_sized(self._ctx, ctypes.byref(t), ctypes.sizeof(t)))
        return dict((f[0], getattr(t, f[0])) for f in L.PcoaTimings._fields_ if not f[0].startswith("reserved"))

    def reset_timings(self):
        self._check(self._lib.pcoa_reset_timings(self._ctx))
