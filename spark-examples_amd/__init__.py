"""spark-examples_amd -- MI355X-native PCoA engine for genomic variants.

The one data-parallel hot path of googlegenomics/spark-examples (VariantsPcaDriver:
getSimilarityMatrix -> computePca) rebuilt from scratch as hand-written HIP kernels for gfx950
behind a C ABI (include/pcoa.h, libpcoa_hip.so).  This package is the Python host over that ABI:

  engine        PcoaEngine: one ctx per GPU
  variants_pca  mirror of the reference driver / Python twin (same function names)
  dist          variant sharding + all-reduce of the partial Gram (torch.distributed, RCCL)
  synth         deterministic synthetic genotypes (bench / tests)
  ingest        local VCF / npz loaders

The directory name contains '-', so import it with
    importlib.import_module("spark-examples_amd")      (tests/conftest.py and bench.py do this).
"""
from . import _lib  # noqa: F401  (binding only; the shared library is loaded on first use)
from .engine import IndexRangeError, PcoaEngine, PcoaError  # noqa: F401

__all__ = ["PcoaEngine", "PcoaError", "IndexRangeError"]
