"""ctypes binding of include/pcoa.h (libpcoa_hip.so).

There is no CPU fallback: if the HIP library is missing this module raises ImportError, and without a
GPU pcoa_create fails with PCOA_ERR_NO_DEVICE.  Nothing here imports oracle/.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PCOA_LIB: another build of the same library (A/B runs of two trees on one box, sanitizer builds); there is no fallback
# either way -- the named file is loaded or the import fails
LIB_PATH = os.environ.get("PCOA_LIB") or os.path.join(_HERE, "libpcoa_hip.so")

PCOA_OK = 0
PCOA_ERR_INVALID_ARG = -1
PCOA_ERR_NO_DEVICE = -2
PCOA_ERR_HIP = -3
PCOA_ERR_OUT_OF_MEMORY = -4
PCOA_ERR_INDEX_RANGE = -5
PCOA_ERR_RCCL = -6
PCOA_ERR_NOT_CONVERGED = -7
PCOA_ERR_STATE = -8

PCOA_FLAG_DEFAULT = 0
PCOA_FLAG_GRAM_F32_MFMA = 0x1
PCOA_FLAG_GRAM_I8_MFMA = 0x2
PCOA_FLAG_GRAM_FP4_MFMA = 0x4
PCOA_FLAG_NO_SIGN_NORM = 0x10
PCOA_FLAG_EIG_HOUSEHOLDER = 0x20
PCOA_FLAG_EIG_LANCZOS = 0x40
PCOA_FLAG_NO_PIPELINE = 0x80
PCOA_FLAG_OPERAND_FP4 = 0x100
PCOA_FLAG_EIG_BAND = 0x200
MATVEC_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p)
PCOA_BED_HOST_ASYNC = 2
PCOA_CALLS_DEVICE_PTR = 1
PCOA_CALLS_HOST_PINNED = 2
PCOA_CALLS_ASYNC = 4


class PcoaTimings(ctypes.Structure):
    _fields_ = [
        ("gram_kernel_seconds", ctypes.c_double),
        ("gram_kernel_launches", ctypes.c_int64),
        ("gram_variants", ctypes.c_int64),
        ("gram_flops", ctypes.c_double),
        ("gram_bytes", ctypes.c_double),
        ("densify_seconds", ctypes.c_double),
        ("synth_seconds", ctypes.c_double),
        ("finalize_seconds", ctypes.c_double),
        ("center_seconds", ctypes.c_double),
        ("tridiag_seconds", ctypes.c_double),
        ("eig_seconds", ctypes.c_double),
        ("backtransform_seconds", ctypes.c_double),
        ("compute_total_seconds", ctypes.c_double),
        ("gram_kernel_kind", ctypes.c_int32),
        ("operand_bits", ctypes.c_int32),
        ("pack_seconds", ctypes.c_double),
        ("pack_launches", ctypes.c_int64),
        ("pack_bytes", ctypes.c_double),
        ("lanczos_seconds", ctypes.c_double),
        ("eig_method", ctypes.c_int32),
        ("lanczos_steps", ctypes.c_int32),
        ("fp4_fallbacks", ctypes.c_int64),
        ("lockstep_launches", ctypes.c_int64),
        ("pipeline_launches", ctypes.c_int64),
        ("pipeline_pre_pass_cus", ctypes.c_int32),
        ("pipeline_contraction_cus", ctypes.c_int32),
        ("evensplit_launches", ctypes.c_int64),
        ("csr_stage_seconds", ctypes.c_double),
        ("csr_wait_seconds", ctypes.c_double),
        ("csr_fast_chunks", ctypes.c_int64),
        ("csr_redo_chunks", ctypes.c_int64),
        ("allreduce_seconds", ctypes.c_double),
        ("allreduce_calls", ctypes.c_int64),
        ("comm_ranks", ctypes.c_int32),
        ("allreduce_int32", ctypes.c_int32),
        ("matvec_form", ctypes.c_int32),
        ("gram_i64_live", ctypes.c_int32),
        ("reduce_int32_calls", ctypes.c_int64),
        ("narrowed_to_int32", ctypes.c_int64),
        ("lanczos_block_steps", ctypes.c_int32),
        ("reserved_r06", ctypes.c_int32),
    ]


class PcoaSynthParams(ctypes.Structure):
    _fields_ = [
        ("seed", ctypes.c_uint64),
        ("n_pops", ctypes.c_int32),
        ("reserved", ctypes.c_int32),
        ("pop_offsets", ctypes.POINTER(ctypes.c_int32)),
        ("thresholds", ctypes.POINTER(ctypes.c_uint32)),
    ]


# every symbol include/pcoa.h declares: (name, restype, argtypes)
_vp = ctypes.c_void_p
_i32 = ctypes.c_int32
_i64 = ctypes.c_int64
_SIGNATURES = [
    ("pcoa_version", ctypes.c_char_p, []),
    ("pcoa_create", ctypes.c_int, [ctypes.POINTER(_vp), _i32, _i32, ctypes.c_uint32]),
    ("pcoa_create_strip", ctypes.c_int, [ctypes.POINTER(_vp), _i32, _i32, _i32, _i32, ctypes.c_uint32]),
    ("pcoa_strip_info", ctypes.c_int, [_vp, ctypes.POINTER(_i32), ctypes.POINTER(_i32)]),
    ("pcoa_strip_col_sums", ctypes.c_int, [_vp, _vp]),
    ("pcoa_strip_matvec", ctypes.c_int, [_vp, _vp, _vp, ctypes.c_double, _vp]),
    ("pcoa_strip_set_centering", ctypes.c_int, [_vp, _vp, ctypes.c_double]),
    ("pcoa_strip_matvec_device", ctypes.c_int, [_vp, _vp, _vp]),
    ("pcoa_destroy", None, [_vp]),
    ("pcoa_last_error", ctypes.c_char_p, [_vp]),
    ("pcoa_reset", ctypes.c_int, [_vp]),
    ("pcoa_n_samples", ctypes.c_int, [_vp]),
    ("pcoa_set_stream", ctypes.c_int, [_vp, _vp]),
    ("pcoa_sync", ctypes.c_int, [_vp]),
    ("pcoa_reserve", ctypes.c_int, [_vp, _i64, _i32]),
    ("pcoa_comm_runtime", ctypes.c_int, [ctypes.c_char_p, _i32, ctypes.POINTER(_i32)]),
    ("pcoa_debug_alloc", ctypes.c_int, [_i32, ctypes.c_size_t, ctypes.POINTER(_vp)]),
    ("pcoa_debug_free", ctypes.c_int, [_vp]),
    ("pcoa_debug_guard_mode", ctypes.c_int, []),
    ("pcoa_accumulate_calls", ctypes.c_int, [_vp, _vp, _vp, _i64]),
    ("pcoa_gram_reduce_from", ctypes.c_int, [_vp, _vp]),
    ("pcoa_lanczos_with_matvec", ctypes.c_int, [_vp, ctypes.c_int32, _vp, _vp, _vp, _vp, ctypes.POINTER(ctypes.c_int32)]),
    ("pcoa_debug_centred_matvec", ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int]),
    ("pcoa_host_alloc_pinned", ctypes.c_int, [ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p)]),
    ("pcoa_host_free_pinned", ctypes.c_int, [_vp]),
    ("pcoa_accumulate_plink_bed", ctypes.c_int, [_vp, _vp, _i64, _i64, ctypes.c_int, ctypes.c_int]),
    ("pcoa_accumulate_calls_ex", ctypes.c_int, [_vp, _vp, _vp, _i64, ctypes.c_uint32]),
    ("pcoa_accumulate_dense_f32", ctypes.c_int, [_vp, _vp, _i64, _i64, ctypes.c_int]),
    ("pcoa_accumulate_dense_u8", ctypes.c_int, [_vp, _vp, _i64, _i64, ctypes.c_int]),
    ("pcoa_accumulate_bits", ctypes.c_int, [_vp, _vp, _i64, _i64, ctypes.c_int]),
    ("pcoa_accumulate_synthetic", ctypes.c_int, [_vp, ctypes.POINTER(PcoaSynthParams), _i64, _i64]),
    ("pcoa_synth_fill_f32", ctypes.c_int, [_vp, ctypes.POINTER(PcoaSynthParams), _i64, _i64, _vp, _i64]),
    ("pcoa_gram_finalize", ctypes.c_int, [_vp]),
    ("pcoa_gram_allreduce_rccl", ctypes.c_int, [_vp, _vp]),
    ("pcoa_comm_unique_id", ctypes.c_int, [_vp]),
    ("pcoa_comm_init", ctypes.c_int, [_vp, _vp, _i32, _i32, ctypes.POINTER(_vp)]),
    ("pcoa_comm_destroy", ctypes.c_int, [_vp]),
    ("pcoa_comm_count", ctypes.c_int, [_vp, ctypes.POINTER(_i32)]),
    ("pcoa_gram_export_device_i64", ctypes.c_int, [_vp, _vp]),
    ("pcoa_gram_import_device_i64", ctypes.c_int, [_vp, _vp]),
    ("pcoa_gram_read_i64", ctypes.c_int, [_vp, _vp]),
    ("pcoa_gram_read_block_i64", ctypes.c_int, [_vp, _i32, _i32, _i32, _i32, _vp]),
    ("pcoa_gram_load_i64", ctypes.c_int, [_vp, _vp]),
    ("pcoa_center_read_f64", ctypes.c_int, [_vp, _vp, _vp, ctypes.POINTER(_i32), ctypes.POINTER(ctypes.c_double)]),
    ("pcoa_compute", ctypes.c_int, [_vp, _i32, _vp, _vp, ctypes.POINTER(_i32)]),
    ("pcoa_get_timings", ctypes.c_int, [_vp, ctypes.POINTER(PcoaTimings)]),
    ("pcoa_get_timings_sized", ctypes.c_int, [_vp, ctypes.POINTER(PcoaTimings), ctypes.c_size_t]),
    ("pcoa_reset_timings", ctypes.c_int, [_vp]),
    ("pcoa_device_info", ctypes.c_int, [_vp, ctypes.c_char_p, _i32, ctypes.POINTER(_i32)]),
]
EXPORTED_SYMBOLS = [s[0] for s in _SIGNATURES]

_lib = None


def source_hash():
    """sha256 (first 16 hex digits) over the kernel sources the library is built from (csrc/* and include/pcoa.h).
    A measurement file made from one tree (profiles/gram_pmc_live.json) names the tree by this hash; .git does not
    travel to the GPU box and the commit that adds such a file moves HEAD, so a commit id cannot be used for it."""
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(_HERE, "csrc")
    files = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".inl", ".h"))]
    files.append(os.path.join(os.path.dirname(_HERE), "include", "pcoa.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load():
    """Loads libpcoa_hip.so and binds every entry point; raises ImportError if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C spark-examples_amd/csrc`.  There is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.  If libpcoa_hip.so pulls in
    # /opt/rocm's copy first and torch is imported later, two runtimes tear down at exit ("double free or
    # corruption").  Importing torch first makes the loader resolve our DT_NEEDED entry to the runtime
    # that is already mapped.  (Plumbing only; skipped when torch is not installed.)
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover
        pass
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, restype, argtypes in _SIGNATURES:
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib
