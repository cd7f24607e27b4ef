"""Host-side mirror of the reference's PCoA driver, on top of the HIP engine.

Same names, argument meaning and error behaviour as the reference interface for this path:

  reference (Scala, VariantsPca.scala)                     here
  ---------------------------------------------------------------------------------------------
  VariantsPcaDriver.extractCallInfo            :56-60      extract_call_info
  VariantsPcaDriver.getCallsRdd                :153-168    VariantsPcaDriver.getCallsRdd / prepare_call_data
  VariantsPcaDriver.getSimilarityMatrix        :182-191    VariantsPcaDriver.getSimilarityMatrix / calculate_similarity_matrix
  VariantsPcaDriver.computePca                 :198-231    VariantsPcaDriver.computePca / center_matrix + perform_pca
  VariantsPcaDriver.emitResult                 :233-246    VariantsPcaDriver.emitResult
  VariantsPcaDriver.main                       :38-50      main
  PcaConf / GenomicsConf flags                 GenomicsConf.scala:31-101   PcaConf

and of the Python twin src/main/python/variants_pca.py (prepare_call_data :19-52,
calculate_similarity_matrix :54-82, center_matrix :84-121, perform_pca :123-152).

An "RDD" here is a plain Python list (of variant dicts, or of per-variant index lists); the
distributed part of the reference (partitions + reduceByKey) is the GPU Gram kernel + all-reduce.
Nothing in this module computes on the CPU: every matrix operation goes through libpcoa_hip.so.
"""
from __future__ import print_function

import argparse
import os
import sys

import numpy as np

if __name__ == "__main__" and not __package__:
    # started by path (`python spark-examples_amd/variants_pca.py ...`; torch.distributed.run starts a rank this way): load the
    # module inside its package -- the relative imports below need one
    import importlib
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.exit(importlib.import_module("spark-examples_amd.variants_pca").main(sys.argv[1:]))

from .engine import PcoaEngine

PLINK_BLOCK_ROWS = 1 << 16   # raw .bed rows handed to pcoa_accumulate_plink_bed per call (41 MB at N = 2504)


# --------------------------------------------------------------------------------------------- a1/a2
def extract_call_info(variant, mapping):
    """VariantsPcaDriver.extractCallInfo (VariantsPca.scala:56-60).

    variant: dict with optional 'calls' -> list of {'callSetId'|'callsetId': str, 'genotype': [int]}.
    Returns [(hasVariation, callsetIndex)]; hasVariation = genotype.exists(_ > 0) -- a missing
    allele (-1) or hom-ref (0) does not vary.  An unknown callset id raises KeyError, as
    mapping(call.callsetId) throws NoSuchElementException in the reference."""
    out = []
    for call in (variant.get("calls") or []):
        cid = call["callSetId"] if "callSetId" in call else call["callsetId"]
        has_variation = False
        for allele in call.get("genotype", []):
            has_variation = has_variation or allele > 0
        out.append((has_variation, mapping[cid]))
    return out


def prepare_call_data(py_rdd, py_id_to_index):
    """prepare_call_data (variants_pca.py:19-52) == getCallsRdd for one variant set
    (VariantsPca.scala:153-157,163-167): keep calls with variation, drop variants with none,
    map callset ids to indices.  Returns a list of index lists (RDD[Seq[Int]]).

    Note: the Python twin tests any(genotype) (non-zero, so -1 counts) where the Scala driver tests
    _ > 0; this mirror follows the Scala driver."""
    call_rdd = []
    for variant in py_rdd:
        calls = [idx for (has_variation, idx) in extract_call_info(variant, py_id_to_index) if has_variation]
        if len(calls) > 0:
            call_rdd.append(calls)
    return call_rdd


# --------------------------------------------------------------------------------------------- join / merge
_M64 = 0xFFFFFFFFFFFFFFFF


def _rotl64(x, r):
    return ((x << r) | (x >> (64 - r))) & _M64


def _fmix64(k):
    k ^= k >> 33
    k = (k * 0xFF51AFD7ED558CCD) & _M64
    k ^= k >> 33
    k = (k * 0xC4CEB9FE1A85EC53) & _M64
    k ^= k >> 33
    return k


def murmur3_128_hex(data, seed=0):
    """Guava Hashing.murmur3_128(seed).hashBytes(data).toString(): MurmurHash3_x64_128, the two 64-bit
    halves printed as little-endian bytes (the reference keys variants with it, VariantsPca.scala:69-77)."""
    c1, c2 = 0x87C37B91114253D5, 0x4CF5AD432745937F
    h1 = h2 = seed & _M64
    n = len(data)
    nblocks = n // 16
    for i in range(nblocks):
        k1 = int.from_bytes(data[16 * i:16 * i + 8], "little")
        k2 = int.from_bytes(data[16 * i + 8:16 * i + 16], "little")
        k1 = (k1 * c1) & _M64; k1 = _rotl64(k1, 31); k1 = (k1 * c2) & _M64; h1 ^= k1
        h1 = _rotl64(h1, 27); h1 = (h1 + h2) & _M64; h1 = (h1 * 5 + 0x52DCE729) & _M64
        k2 = (k2 * c2) & _M64; k2 = _rotl64(k2, 33); k2 = (k2 * c1) & _M64; h2 ^= k2
        h2 = _rotl64(h2, 31); h2 = (h2 + h1) & _M64; h2 = (h2 * 5 + 0x38495AB5) & _M64
    tail = data[16 * nblocks:]
    k1 = k2 = 0
    if len(tail) > 8:
        k2 = int.from_bytes(tail[8:], "little")
        k2 = (k2 * c2) & _M64; k2 = _rotl64(k2, 33); k2 = (k2 * c1) & _M64; h2 ^= k2
    if len(tail) > 0:
        k1 = int.from_bytes(tail[:8], "little")
        k1 = (k1 * c1) & _M64; k1 = _rotl64(k1, 31); k1 = (k1 * c2) & _M64; h1 ^= k1
    h1 ^= n; h2 ^= n
    h1 = (h1 + h2) & _M64; h2 = (h2 + h1) & _M64
    h1 = _fmix64(h1); h2 = _fmix64(h2)
    h1 = (h1 + h2) & _M64; h2 = (h2 + h1) & _M64
    return (h1.to_bytes(8, "little") + h2.to_bytes(8, "little")).hex()


def get_variant_key(variant, debug=False):
    """VariantsPcaDriver.getVariantKey (VariantsPca.scala:62-78): murmur3_128 over
    contig, start, end, referenceBases, alternateBases.mkString("") as a Guava Hasher sees them
    (putString = UTF-8 bytes, putLong = 8 little-endian bytes)."""
    alternate = "".join(variant.get("alternateBases") or [])
    reference = variant.get("referenceBases") or ""
    if debug:
        print("%s: (%d, %d) ref=%s alt=%s" % (variant["contig"], variant["start"], variant["end"], reference, alternate))
    buf = (variant["contig"].encode("utf-8") + int(variant["start"]).to_bytes(8, "little", signed=True) +
           int(variant["end"]).to_bytes(8, "little", signed=True) + reference.encode("utf-8") +
           alternate.encode("utf-8"))
    return murmur3_128_hex(buf)


def join_datasets(datasets, indexes, debug=False):
    """VariantsPcaDriver.joinDatasets (VariantsPca.scala:115-128): two-way INNER join on the variant
    key; the joined record is calls1 ++ calls2 (cross product if a key repeats, as RDD.join does)."""
    keyed = []
    for data in datasets[:2]:
        d = {}
        for variant in data:
            d.setdefault(get_variant_key(variant, debug), []).append(extract_call_info(variant, indexes))
        keyed.append(d)
    out = []
    for key, calls1 in keyed[0].items():
        for c1 in calls1:
            for c2 in keyed[1].get(key, []):
                out.append(c1 + c2)
    return out


def merge_datasets(datasets, variant_set_count, indexes):
    """VariantsPcaDriver.mergeDatasets (VariantsPca.scala:136-148): union, group by variant key, keep
    the groups with exactly variantSetCount members, concatenate their calls."""
    groups = {}
    for data in datasets:
        for variant in data:
            groups.setdefault(get_variant_key(variant), []).append(extract_call_info(variant, indexes))
    return [[c for calls in g for c in calls] for g in groups.values() if len(g) == variant_set_count]


# --------------------------------------------------------------------------------------------- a3
def calculate_similarity_matrix(call_rdd, matrix_size, engine=None, device=0):
    """calculate_similarity_matrix (variants_pca.py:54-82) == getSimilarityMatrix
    (VariantsPca.scala:182-191).  call_rdd: list of index lists, or a (sample_idx, row_offsets) CSR
    pair.  Returns the PcoaEngine holding S in HBM (use .gram() for the N^2 entries)."""
    eng = engine if engine is not None else PcoaEngine(matrix_size, device=device)
    if isinstance(call_rdd, tuple) and isinstance(call_rdd[0], str) and call_rdd[0] == "bed":
        # a PLINK fileset as it lies in the file: blocks of raw 2-bit rows, decoded on the device (pcoa_accumulate_plink_bed);
        # rows outside --references are squeezed out of a block before it is handed over
        _, geno, keep, ref_is_a1 = call_rdd
        all_kept = bool(keep.all())
        for v0 in range(0, geno.shape[0], PLINK_BLOCK_ROWS):
            rows = geno[v0:v0 + PLINK_BLOCK_ROWS]
            if not all_kept:
                k = keep[v0:v0 + PLINK_BLOCK_ROWS]
                if not k.any():
                    continue
                rows = rows[k]
            eng.accumulate_plink_bed(np.ascontiguousarray(rows), ref_is_a1=ref_is_a1)
    elif isinstance(call_rdd, tuple) and isinstance(call_rdd[0], str) and call_rdd[0] == "bits":
        bits = call_rdd[1]                          # carrier bitsets [variants][ceil(N / 32)] (a PLINK fileset)
        for v0 in range(0, bits.shape[0], 1 << 20):
            eng.accumulate_bits(bits[v0:v0 + (1 << 20)])
    elif isinstance(call_rdd, tuple):
        eng.accumulate_calls(call_rdd[0], call_rdd[1])
    else:
        eng.accumulate_callsets(call_rdd)
    eng.finalize()
    return eng


def shard_calls(call_rdd, rank, world):
    """Rank `rank`'s partition of an RDD[Seq[Int]] in any of the forms getCallsRdd returns -- the contiguous range
    dist.shard_range(rank, world, rows), as the reference's partitions are contiguous ranges of the variant stream
    (VariantsPca.scala:184; numReducePartitions, GenomicsConf.scala:42-45).  The shards of all ranks concatenate to the
    input; S = sum over variants, so the all-reduce of the ranks' partial matrices is the reference's reduceByKey (:190)."""
    from . import dist
    if isinstance(call_rdd, tuple) and isinstance(call_rdd[0], str) and call_rdd[0] == "bed":
        _, geno, keep, ref_is_a1 = call_rdd
        a, b = dist.shard_range(rank, world, geno.shape[0])
        return ("bed", geno[a:b], keep[a:b], ref_is_a1)
    if isinstance(call_rdd, tuple) and isinstance(call_rdd[0], str) and call_rdd[0] == "bits":
        a, b = dist.shard_range(rank, world, call_rdd[1].shape[0])
        return ("bits", call_rdd[1][a:b])
    if isinstance(call_rdd, tuple):
        idx, offs = np.asarray(call_rdd[0]), np.asarray(call_rdd[1])
        a, b = dist.shard_range(rank, world, offs.size - 1)
        return (idx[offs[a]:offs[b]], offs[a:b + 1] - offs[a])
    a, b = dist.shard_range(rank, world, len(call_rdd))
    return call_rdd[a:b]


# --------------------------------------------------------------------------------------------- a5/a6
def center_matrix(sim_matrix, row_count):
    """center_matrix (variants_pca.py:84-121; VariantsPca.scala:199-223).
    sim_matrix: PcoaEngine (as returned above) or an N x N integer array.  Returns B (N x N fp64)."""
    eng = _as_engine(sim_matrix, row_count)
    b, _, _, _ = eng.center()
    return b


# --------------------------------------------------------------------------------------------- a7
def perform_pca(matrix, row_count, nr_principal_components=2):
    """perform_pca (variants_pca.py:123-152; VariantsPca.scala:224-227).  `matrix` is the engine
    holding S (centring is part of computePca on the device) or an N x N similarity array.
    Returns an N x k numpy array (unit-norm columns, sign-normalised)."""
    eng = _as_engine(matrix, row_count)
    comps, _, _ = eng.compute(nr_principal_components)
    return comps


def _as_engine(obj, row_count):
    if isinstance(obj, PcoaEngine):
        if obj.n != row_count:
            raise ValueError("row_count %d does not match the engine's N = %d" % (row_count, obj.n))
        return obj
    eng = PcoaEngine(row_count)
    eng.load_gram(np.asarray(obj))
    return eng


# --------------------------------------------------------------------------------------------- conf
class PcaConf(object):
    """Flags of PcaConf / GenomicsConf (GenomicsConf.scala:31-101), same names and defaults.
    Spark- and cloud-specific flags are accepted and ignored; --input-path names a local dataset
    (the Genomics API the reference streamed from no longer exists)."""

    def __init__(self, arguments):
        p = argparse.ArgumentParser(prog="VariantsPcaDriver", description=self.__doc__)
        p.add_argument("--bases-per-partition", type=int, default=1000000)
        p.add_argument("--client-secrets", type=str, default=None)
        p.add_argument("--input-path", type=str, nargs="*", default=None,
                       help="one local dataset per variant set: .npz (callset_ids, [callset_names], sample_idx, "
                            "row_offsets) or .vcf[.gz]; two files are joined, three or more merged "
                            "(VariantsPca.scala:153-162)")
        p.add_argument("--num-reduce-partitions", type=int, default=10)
        p.add_argument("--output-path", type=str, default=None)
        p.add_argument("--references", type=str, nargs="*", default=["chr17:41196311:41277499"])
        p.add_argument("--spark-master", type=str, default=None)
        p.add_argument("--variant-set-id", type=str, nargs="*", default=["3049512673186936334"])
        p.add_argument("--all-references", action="store_true")
        p.add_argument("--debug-datasets", action="store_true")
        p.add_argument("--min-allele-frequency", type=float, default=None)
        p.add_argument("--num-pc", type=int, default=2)
        # additions of this engine
        p.add_argument("--synthetic", type=str, default=None, help="V,N,seed: synthetic Balding-Nichols input")
        p.add_argument("--gpu", type=int, default=0)
        p.add_argument("--gpus", type=int, default=1,
                       help="K > 1: one process per GPU (started here through torch.distributed.run unless a launcher already "
                            "did), the variants partitioned into K contiguous shards, one RCCL all-reduce of the partial "
                            "similarity matrices (reduceByKey, VariantsPca.scala:190), computePca and the output on rank 0")
        p.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                       help="--gpus K: torch.distributed backend of the rendezvous (nccl = RCCL; gloo = CPU wire, for boxes where "
                            "several ranks have to share one GPU -- it implies --allreduce torch)")
        p.add_argument("--rank-devices", type=str, default=None,
                       help="--gpus K: device ordinal of each rank, comma-separated (default: rank r on GPU r); ordinals may repeat "
                            "with --dist-backend gloo (how the path is tested on a one-GPU box)")
        p.add_argument("--allreduce", choices=["native", "torch"], default="native",
                       help="--gpus K: native = the library's RCCL communicator (in place on the int32 partial), torch = "
                            "export -> torch.distributed.all_reduce -> import")
        p.add_argument("--plink-ref-allele", choices=["a1", "a2"], default="a2",
                       help="PLINK filesets: which .bim allele column is the reference allele (a2: written with "
                            "--keep-allele-order / plink2 --make-bed; a1: the other way round)")
        p.add_argument("--spark-output-layout", action="store_true",
                       help="write <output-path>-pca.tsv as the DIRECTORY Spark's saveAsTextFile leaves (part-00000 + _SUCCESS, "
                            "VariantsPca.scala:241-245) instead of one file of that name")
        p.add_argument("--dump-similarity", type=str, default=None,
                       help="write S (N x N int64, little-endian, row-major) to this file (parity tests)")
        a = p.parse_args(list(arguments))
        self.__dict__.update(vars(a))
        self.numPc = a.num_pc
        self.outputPath = a.output_path
        self.inputPath = a.input_path
        self.minAlleleFrequency = a.min_allele_frequency


def java_float_to_string(f):
    """Float.toString (the reference prints --min-allele-frequency, a Float, at VariantsPca.scala:99)."""
    f32 = np.float32(f)
    return java_double_to_string(float(f32), shortest=np.format_float_scientific(abs(f32), unique=True, trim="0"))


def java_double_to_string(d, shortest=None):
    """Double.toString, which the reference's string interpolation uses (VariantsPca.scala:239):
    decimal for 1e-3 <= |d| < 1e7, otherwise computerised scientific notation (1.0E-4).
    `shortest`: the shortest round-trip digits of |d| when they are not repr(double)'s (Float.toString)."""
    d = float(d)
    if d != d:
        return "NaN"
    if d in (float("inf"), float("-inf")):
        return "Infinity" if d > 0 else "-Infinity"
    if d == 0.0:
        return "-0.0" if str(d).startswith("-") else "0.0"
    r = shortest if shortest is not None else repr(abs(d))
    mant, _, exp = r.partition("e")
    e10 = int(exp) if exp else 0
    ip, _, fp = mant.partition(".")
    digits = (ip + fp).lstrip("0")
    # decimal exponent of the first significant digit
    if ip.strip("0"):
        e10 += len(ip.lstrip("0")) - 1
    else:
        e10 -= len(fp) - len(fp.lstrip("0")) + 1
    digits = digits.rstrip("0") or "0"
    sign = "-" if d < 0 else ""
    if 1e-3 <= abs(d) < 1e7:
        if e10 >= 0:
            whole = digits[:e10 + 1].ljust(e10 + 1, "0")
            frac = digits[e10 + 1:] or "0"
        else:
            whole = "0"
            frac = "0" * (-e10 - 1) + digits
        return "%s%s.%s" % (sign, whole, frac)
    return "%s%s.%sE%d" % (sign, digits[0], digits[1:] or "0", e10)


# --------------------------------------------------------------------------------------------- driver
def similarity_entries_nonzero(s):
    """RDD[((Int, Int), Int)] of getSimilarityMatrixStream (VariantsPca.scala:262-279) from a dense S: the keys whose
    count is non-zero, both triangles (the reference emits the pairs c1 <= c2 and mirrors the strict upper ones)."""
    s = np.asarray(s)
    rows, cols = np.nonzero(s)
    return [((int(i), int(j)), int(s[i, j])) for i, j in zip(rows, cols)]


class VariantsPcaDriver(object):
    """class VariantsPcaDriver (VariantsPca.scala:81-286) over a local dataset.

    indexes: callset id -> 0..N-1 in callset-list order, names: callset id -> name
    (VariantsCommon.scala:44-47; the ordering rule is preserved)."""

    def __init__(self, conf, indexes, names, data):
        self.conf = conf
        self.indexes = dict(indexes)
        self.names = dict(names)
        self.data = data  # list of datasets, each a list of variant dicts (or ('csr', idx, offs))
        print("Matrix size: %d." % len(self.indexes))  # VariantsCommon.scala:48
        self.engine = None

    # filterDataset, VariantsPca.scala:96-108
    def filterDataset(self, data):
        maf = self.conf.minAlleleFrequency
        if maf is None:
            return data
        if isinstance(data, tuple):
            # carrier-only inputs (.npz CSR, --synthetic) carry no INFO/AF: the filter cannot be applied, and ignoring
            # it silently would change the result
            raise ValueError("--min-allele-frequency needs variant records with INFO/AF (a VCF input), "
                             "not pre-extracted carriers (.npz / --synthetic)")
        print("Min allele frequency %s." % java_float_to_string(maf))
        out = []
        for variant in data:
            af = (variant.get("info") or {}).get("AF")
            if af is not None and len(af) > 0 and np.float32(af[0]) >= np.float32(maf):
                out.append(variant)
        return out

    # getCallsRdd, VariantsPca.scala:153-168
    def getCallsRdd(self, data):
        variant_set_count = len(data)
        if variant_set_count == 1:
            d = data[0]
            if isinstance(d, tuple):  # pre-extracted carriers: CSR rows (already filtered) or bitsets
                if d[0] == "bed":
                    return d
                return ("bits", d[1]) if d[0] == "bits" else (d[1], d[2])
            return prepare_call_data(d, self.indexes)
        if any(isinstance(d, tuple) for d in data):
            raise ValueError("joining datasets needs variant records (contig/start/end/ref/alt), not CSR carriers")
        if variant_set_count == 2:
            callsets = join_datasets(data, self.indexes, self.conf.debug_datasets)
        else:
            callsets = merge_datasets(data, variant_set_count, self.indexes)
        out = []
        for calls in callsets:  # :164-167
            kept = [idx for (has_variation, idx) in calls if has_variation]
            if len(kept) > 0:
                out.append(kept)
        return out

    # getSimilarityMatrix, VariantsPca.scala:182-191
    def getSimilarityMatrix(self, callsets):
        self.engine = calculate_similarity_matrix(callsets, len(self.indexes), device=self.conf.gpu)
        return self.engine

    # the same with the reference's parallel strategy (:184-190): this rank's partition of the variants on this rank's GPU,
    # then the sum of the partial matrices over the ranks (reduceByKey(_ + _, numReducePartitions)).  Every rank ends up
    # with the whole S.  Returns (engine, telemetry).
    def getSimilarityMatrixSharded(self, callsets, rank, world, device, allreduce="native"):
        import time
        from . import dist
        shard = shard_calls(callsets, rank, world)
        t0 = time.perf_counter()
        self.engine = calculate_similarity_matrix(shard, len(self.indexes), device=device)
        self.engine.sync()
        t_local = time.perf_counter() - t0
        native = None
        if allreduce == "native":
            native = dist.NativeComm(self.engine)
        t1 = time.perf_counter()
        if native is not None:
            native.allreduce()
            self.engine.sync()
        else:
            dist.allreduce_engine(self.engine)
        t_red = time.perf_counter() - t1
        tele = dist.collective_telemetry(t_local, t_red, self.engine.timings(), native.count() if native is not None else None)
        if native is not None:
            native.close()
        return self.engine, tele

    # getSimilarityMatrixStream, VariantsPca.scala:262-279 (not called by the reference's main): the same S through
    # upper-triangle pair emission + mirror, i.e. only the keys with a non-zero count exist
    def getSimilarityMatrixStream(self, callsets):
        self.engine = calculate_similarity_matrix(callsets, len(self.indexes), device=self.conf.gpu)
        return similarity_entries_nonzero(self.engine.gram())

    # computePca, VariantsPca.scala:198-231
    def computePca(self, sim_matrix):
        n = len(self.indexes)
        comps, _, nonzero = sim_matrix.compute(self.conf.numPc)
        print("Non zero rows in matrix: %d / %d." % (nonzero, n))  # :208
        if comps.shape[1] < 2:
            # the reference indexes array(i + pca.numRows) unconditionally (:230) and fails for --num-pc 1
            raise IndexError("computePca emits exactly PC1 and PC2 (VariantsPca.scala:229-230); --num-pc must be >= 2")
        reverse = dict((v, k) for (k, v) in self.indexes.items())
        return [(reverse[i], float(comps[i, 0]), float(comps[i, 1])) for i in range(n)]

    # emitResult, VariantsPca.scala:233-246.  stdout: name, dataset, pc1, pc2 sorted by name, as the reference prints.
    # File: the reference hands the rows to Spark's saveAsTextFile, i.e. a DIRECTORY <output-path>-pca.tsv/ of unsorted
    # part-* files (name, pc1, pc2, dataset per line); here the same lines go, sorted by name, into ONE file of that name.
    def emitResult(self, result, out=None):
        out = out or sys.stdout
        rows = []
        for (callset_id, pc1, pc2) in result:
            dataset = callset_id.split("-")[0]
            rows.append((self.names[callset_id], pc1, pc2, dataset))
        rows.sort(key=lambda t: t[0])
        for (name, pc1, pc2, dataset) in rows:
            out.write("%s\t%s\t%s\t%s\n" % (name, dataset, java_double_to_string(pc1), java_double_to_string(pc2)))
        if self.conf.outputPath:
            target = self.conf.outputPath + "-pca.tsv"
            if getattr(self.conf, "spark_output_layout", False):   # saveAsTextFile (:241-245): a directory; an existing one is an error
                os.mkdir(target)
                open(os.path.join(target, "_SUCCESS"), "w").close()
                target = os.path.join(target, "part-00000")
            with open(target, "w") as f:
                for (name, pc1, pc2, dataset) in rows:
                    f.write("%s\t%s\t%s\t%s\n" % (name, java_double_to_string(pc1), java_double_to_string(pc2), dataset))

    def reportIoStats(self, out=None):
        out = out or sys.stdout
        if self.engine is not None:
            t = self.engine.timings()
            out.write("Variants accumulated: %d; Gram kernel %.3f ms; PCoA %.3f ms\n" %
                      (t["gram_variants"], 1e3 * t["gram_kernel_seconds"], 1e3 * t["compute_total_seconds"]))

    def stop(self):
        if self.engine is not None:
            self.engine.close()
            self.engine = None


def load_dataset(conf):
    """Local stand-in for VariantsCommon (VariantsCommon.scala:33-66): callset index/name maps +
    the variant data.  Returns (indexes, names, [dataset])."""
    from . import ingest
    if conf.synthetic:
        return ingest.synthetic_dataset(conf.synthetic)
    if not conf.inputPath:
        raise SystemExit("--input-path (local .vcf[.gz], PLINK .bed/.bim/.fam or .npz) or --synthetic V,N,seed is required: "
                         "the Google Genomics API the reference read from has been shut down")
    paths = conf.inputPath if isinstance(conf.inputPath, (list, tuple)) else [conf.inputPath]
    refs = None if conf.all_references else conf.references
    if len(paths) == 1 and conf.minAlleleFrequency is None:
        if paths[0].endswith(".npz"):
            return ingest.load_npz(paths[0])
        if paths[0][-4:] in (".bed", ".bim", ".fam"):
            return ingest.load_plink(paths[0], refs, ref_allele=conf.plink_ref_allele, as_bed=True)
        return ingest.load_vcf(paths[0], refs)
    # several variant sets (or the AF filter): full variant records are needed for keys and INFO/AF
    if any(p.endswith(".npz") or p[-4:] in (".bed", ".bim", ".fam") for p in paths):
        raise SystemExit("joining variant sets or filtering by allele frequency needs VCF inputs: a .npz dataset or a PLINK "
                         "fileset is read as carriers only (no contig/start/end/ref/alt keys, no INFO/AF)")
    print("Running PCA on %d datasets." % len(paths))  # VariantsCommon.scala:57
    indexes, names, data = {}, {}, []
    used = set()
    for k, path in enumerate(paths):
        set_id = ingest.set_id_of(path, k, used)
        ids, nm, variants = ingest.load_vcf_records(path, parse_refs(refs, k), set_id=set_id)
        # callset index = position in the concatenated callset lists (VariantsCommon.scala:44-45), as the compiled
        # host assigns them; the ids are unique by construction (set_id_of), so nothing is overwritten
        base = len(indexes)
        for i, cid in enumerate(ids):
            if cid in indexes:
                raise ValueError("duplicate callset id %r" % cid)
            indexes[cid] = base + i
        names.update(nm)
        data.append(variants)
    return indexes, names, data


def parse_refs(refs, k):
    """--references holds one list of tuples per variant set, in order (GenomicsConf.scala:47-51)."""
    if not refs:
        return None
    return [refs[k]] if k < len(refs) else [refs[-1]]


def multi_gpu_plan(conf, args, env, device_count, python=None):
    """What `--gpus K` makes of this invocation (dist.launch_plan): ("run", None) = this process is a rank (or K == 1),
    ("spawn", command) = re-execute under torch.distributed.run with K ranks, ("error", message)."""
    from . import dist
    if conf.rank_devices:   # an explicit map: what has to exist is the highest ordinal it names, not K devices
        devs = [int(t) for t in conf.rank_devices.split(",")]
        if len(devs) != conf.gpus:
            return "error", "--rank-devices must name exactly --gpus devices"
        if len(set(devs)) < len(devs) and conf.dist_backend != "gloo":
            return "error", "--rank-devices repeats a device: RCCL needs one GPU per rank (use --dist-backend gloo on a test box)"
        if max(devs) >= device_count or min(devs) < 0:
            return "error", "--rank-devices names GPU %d but only %d GPU(s) are visible" % (max(devs), device_count)
        device_count = max(device_count, conf.gpus)
    return dist.launch_plan(conf.gpus, env, device_count, [os.path.abspath(__file__)] + list(args), python=python or sys.executable)


def main(args):
    """VariantsPcaDriver.main (VariantsPca.scala:38-50)."""
    conf = PcaConf(args)
    rank, world = 0, 1
    if conf.gpus > 1 or "WORLD_SIZE" in os.environ:
        import torch  # plumbing: rendezvous and the device count
        what, detail = multi_gpu_plan(conf, args, os.environ, torch.cuda.device_count())
        if what == "error":
            raise SystemExit("VariantsPcaDriver: " + detail)
        if what == "spawn":
            import subprocess
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            return subprocess.call(detail, env=env)
        world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch
        import torch.distributed as td
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
        if conf.rank_devices:
            local_rank = [int(t) for t in conf.rank_devices.split(",")][rank]
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if conf.dist_backend == "gloo":
            conf.allreduce = "torch"   # the library's communicator is RCCL
            td.init_process_group("gloo", rank=rank, world_size=world)
        else:
            td.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        quiet = open(os.devnull, "w") if rank != 0 else None   # the reference's driver prints once
        if quiet is not None:
            sys.stdout = quiet
    indexes, names, data = load_dataset(conf)
    driver = VariantsPcaDriver(conf, indexes, names, data)
    filtered = [driver.filterDataset(d) for d in driver.data]
    calls_rdd = driver.getCallsRdd(filtered)
    if world > 1:
        sim_matrix, tele = driver.getSimilarityMatrixSharded(calls_rdd, rank, world, local_rank, conf.allreduce)
        if rank == 0:
            sys.stderr.write("Reduced over %d ranks (RCCL communicator: %s ranks): all-reduce %.3f ms; per-rank accumulate "
                             "%.3f .. %.3f s\n" % (world, tele["rccl_ranks"], tele["allreduce_ms"], tele["rank_elapsed_min_s"],
                                                   tele["rank_elapsed_max_s"]))
    else:
        sim_matrix = driver.getSimilarityMatrix(calls_rdd)
    if rank == 0:
        if conf.dump_similarity:
            sim_matrix.gram().astype("<i8").tofile(conf.dump_similarity)
        result = driver.computePca(sim_matrix)
        driver.emitResult(result)
        driver.reportIoStats(sys.stderr)
    if world > 1:
        import torch.distributed as td
        td.barrier()
        td.destroy_process_group()
    driver.stop()
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
