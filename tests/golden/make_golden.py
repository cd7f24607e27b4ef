#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own Python code in this container.

The reference twin /root/reference/src/main/python/variants_pca.py is Python-2 + PySpark and
cannot be imported as is.  This script
  1. reads it from /root/reference (nothing is copied into the repo),
  2. translates it in memory with lib2to3 (print statement, tuple-parameter lambdas, xrange),
  3. drops its trailing top-level call ``pca(sys.argv[1:])`` (needs a JVM),
  4. executes it against a ~100-line in-memory stand-in for the few pyspark RDD operations it uses
     (map, filter, mapPartitions, reduceByKey, groupByKey, sortByKey, cache, collect, broadcast),
  5. runs the reference functions prepare_call_data (:19-52), calculate_similarity_matrix (:54-82)
     and center_matrix (:84-121) on seeded inputs and stores inputs + outputs as tests/golden/*.npz.

perform_pca (:123-152) only forwards to the JVM (Spark MLlib RowMatrix); it cannot run without a
JVM, so the PCA stage is NOT pinned by these fixtures (the oracle restates MLlib's algorithm).

/root/reference does not exist on the GPU box: this script runs here only; tests read the .npz.
Usage:  python tests/golden/make_golden.py
"""
import json
import os
import sys
import types
import warnings

import numpy as np

REF = "/root/reference/src/main/python/variants_pca.py"
OUT = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------- mock pyspark
class MiniRDD(object):
    """Partitioned in-memory list with the handful of RDD methods variants_pca.py calls."""

    def __init__(self, partitions):
        self.partitions = [list(p) for p in partitions]

    def map(self, f):
        return MiniRDD([[f(x) for x in p] for p in self.partitions])

    def filter(self, f):
        return MiniRDD([[x for x in p if f(x)] for p in self.partitions])

    def mapPartitions(self, f):
        return MiniRDD([list(f(iter(p))) for p in self.partitions])

    def _shuffle(self, nparts):
        buckets = [dict() for _ in range(nparts)]
        for p in self.partitions:
            for k, v in p:
                buckets[hash(k) % nparts].setdefault(k, []).append(v)
        return buckets

    def reduceByKey(self, f, numPartitions=None):
        nparts = numPartitions or max(1, len(self.partitions))
        out = []
        for b in self._shuffle(nparts):
            part = []
            for k, vs in b.items():
                acc = vs[0]
                for v in vs[1:]:
                    acc = f(acc, v)
                part.append((k, acc))
            out.append(part)
        return MiniRDD(out)

    def groupByKey(self, numPartitions=None):
        nparts = numPartitions or max(1, len(self.partitions))
        return MiniRDD([[(k, list(vs)) for k, vs in b.items()] for b in self._shuffle(nparts)])

    def sortByKey(self, ascending=True):
        allkv = [kv for p in self.partitions for kv in p]
        allkv.sort(key=lambda kv: kv[0], reverse=not ascending)
        n = max(1, len(self.partitions))
        step = (len(allkv) + n - 1) // n if allkv else 1
        return MiniRDD([allkv[i:i + step] for i in range(0, max(len(allkv), 1), step)])

    def cache(self):
        return self

    def collect(self):
        return [x for p in self.partitions for x in p]


class _Broadcast(object):
    def __init__(self, v):
        self.value = v


class _SparkContext(object):
    _active_spark_context = None

    def __init__(self, conf=None):
        _SparkContext._active_spark_context = self

    def broadcast(self, v):
        return _Broadcast(v)

    def parallelize(self, data, numSlices=1):
        data = list(data)
        n = max(1, numSlices)
        bounds = [len(data) * i // n for i in range(n + 1)]
        return MiniRDD([data[bounds[i]:bounds[i + 1]] for i in range(n)])


def _install_mock_pyspark():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m
    pyspark = mod("pyspark")
    pyspark.SparkContext = _SparkContext
    pyspark.serializers = mod("pyspark.serializers")
    pyspark.conf = mod("pyspark.conf")
    pyspark.conf.SparkConf = lambda: object()
    pyspark.mllib = mod("pyspark.mllib")
    pyspark.mllib.common = mod("pyspark.mllib.common")
    pyspark.mllib.linalg = mod("pyspark.mllib.linalg")
    pyspark.rdd = mod("pyspark.rdd")
    return pyspark


def load_reference_module():
    """Translate + exec the reference twin; returns its namespace dict."""
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        from lib2to3 import refactor
    fixers = refactor.get_fixers_from_package("lib2to3.fixes")
    tool = refactor.RefactoringTool(fixers)
    src = open(REF).read()
    tail = "pca(sys.argv[1:])"
    assert src.rstrip().endswith(tail), "reference layout changed"
    src = src.rstrip()[:-len(tail)] + "\n"
    py3 = str(tool.refactor_string(src, "variants_pca.py"))
    _install_mock_pyspark()
    if not hasattr(np, "int"):
        np.int = int  # numpy.int (variants_pca.py:68) was removed in numpy 1.24
    ns = {"__name__": "reference_variants_pca"}
    exec(compile(py3, REF, "exec"), ns)
    return ns


# ----------------------------------------------------------------------------- inputs
def make_variants(rng, n_samples, n_variants, pop_freq=None, with_edges=True):
    """Variant records shaped like the dicts prepare_call_data consumes
    ({'calls': [{'callSetId': str, 'genotype': [a, b]}, ...]})."""
    ids = ["set%d-%03d" % (i % 2, i) for i in range(n_samples)]
    variants = []
    for v in range(n_variants):
        if pop_freq is None:
            p = np.full(n_samples, rng.uniform(0.02, 0.6))
        else:
            p = pop_freq[rng.integers(0, pop_freq.shape[0])]
        calls = []
        for i in range(n_samples):
            if with_edges and rng.random() < 0.05:
                continue  # sample has no call at this site (ragged)
            g = [int(rng.random() < p[i]), int(rng.random() < p[i])]
            if with_edges and rng.random() < 0.03:
                g = [int(rng.integers(1, 3))]  # haploid call, allele index may be 2
            calls.append({"callSetId": ids[i], "genotype": g})
        if with_edges and v % 17 == 5:
            variants.append({})  # variant with no 'calls' key at all
        elif with_edges and v % 19 == 7:
            variants.append({"calls": [{"callSetId": c["callSetId"], "genotype": [0, 0]}
                                       for c in calls]})  # nobody varies -> row dropped
        else:
            variants.append({"calls": calls})
    return ids, variants


def run_case(ns, name, ids, variants, n_partitions):
    sc = _SparkContext._active_spark_context or _SparkContext()
    id_to_index = dict((cid, i) for i, cid in enumerate(ids))
    n = len(ids)
    py_rdd = sc.parallelize(variants, n_partitions)
    call_rdd = ns["prepare_call_data"](py_rdd, id_to_index)
    callsets = call_rdd.collect()
    sim = ns["calculate_similarity_matrix"](call_rdd, n)
    entries = sim.collect()
    s = np.zeros((n, n), dtype=np.int64)
    seen = np.zeros((n, n), dtype=bool)
    for (y, x), v in entries:
        assert not seen[y, x]
        seen[y, x] = True
        s[y, x] = int(v)
    assert seen.all(), "reference emits all N^2 keys (variants_pca.py:73-75)"
    centered = ns["center_matrix"](sim, n).collect()
    assert len(centered) == n
    b = np.zeros((n, n), dtype=np.float64)
    for row, colvals in enumerate(centered):  # rows are sorted by key (sortByKey(True), :102)
        for col, val in colvals:
            b[row, col] = val
    offs = np.zeros(len(callsets) + 1, dtype=np.int64)
    for v, c in enumerate(callsets):
        offs[v + 1] = offs[v] + len(c)
    idx = np.array([i for c in callsets for i in c], dtype=np.int32)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, n_samples=np.int64(n), callset_ids=np.array(ids),
                        variants_json=np.array(json.dumps(variants)),
                        sample_idx=idx, row_offsets=offs, similarity=s, centered=b,
                        n_partitions=np.int64(n_partitions))
    print("wrote %s: N=%d, variants in=%d, kept=%d, sum(S)=%d" %
          (path, n, len(variants), len(callsets), int(s.sum())))


def main():
    ns = load_reference_module()
    # 1. the hand-checkable known-answer case of SURVEY.md section 8c
    ids = ["kat-%d" % i for i in range(5)]
    kat = [[0, 1], [0, 1, 2], [3, 4], [2, 3, 4], [0], [1, 4]]
    variants = [{"calls": [{"callSetId": ids[i], "genotype": [0, 1] if i in c else [0, 0]}
                           for i in range(5)]} for c in kat]
    run_case(ns, "kat5", ids, variants, 2)
    # 2. ragged random case, 3 partitions
    rng = np.random.default_rng(20260921)
    ids, variants = make_variants(rng, 16, 64)
    run_case(ns, "ragged16", ids, variants, 3)
    # 3. planted 3-population structure, 4 partitions
    rng = np.random.default_rng(7)
    pops = np.repeat(np.arange(3), [14, 13, 13])
    freqs = np.stack([np.where(pops == k, 0.55, 0.08) for k in range(3)] +
                     [np.full(40, 0.2), np.full(40, 0.02)])
    ids, variants = make_variants(rng, 40, 200, pop_freq=freqs)
    run_case(ns, "pops40", ids, variants, 4)
    # 4. a sample that never varies (all-zero row/column in S) and a single-variant input
    rng = np.random.default_rng(11)
    ids, variants = make_variants(rng, 9, 30, with_edges=False)
    for var in variants:
        for c in var["calls"]:
            if c["callSetId"] == ids[4]:
                c["genotype"] = [0, 0]
    run_case(ns, "zerorow9", ids, variants, 2)
    ids, variants = make_variants(np.random.default_rng(3), 6, 1, with_edges=False)
    variants[0]["calls"][2]["genotype"] = [1, 0]
    run_case(ns, "single6", ids, variants, 1)
    # 5. wider than one 64-sample MFMA column block, five partitions, ragged records, a triploid call
    rng = np.random.default_rng(70)
    ids, variants = make_variants(rng, 70, 300)
    variants[3]["calls"][0]["genotype"] = [0, 2, 1]
    run_case(ns, "wide70", ids, variants, 5)
    # 6. dense carriers, a single partition, N just over a 32-sample MFMA tile
    rng = np.random.default_rng(33)
    ids, variants = make_variants(rng, 33, 400, pop_freq=np.stack([np.full(33, 0.8), np.full(33, 0.45)]),
                                  with_edges=False)
    run_case(ns, "dense33", ids, variants, 1)
    # 7. / 8. tile-edge cases: N crosses the 128- and 256-sample tile edges of the Gram kernels,
    # the variant count crosses k-block / stage edges (32-, 128-variant) and the split-K path; >= 3 partitions
    rng = np.random.default_rng(260)
    pops = np.repeat(np.arange(4), [70, 65, 65, 60])
    freqs = np.stack([np.where(pops == k, 0.5, 0.06) for k in range(4)] + [np.full(260, 0.15)])
    ids, variants = make_variants(rng, 260, 600, pop_freq=freqs)
    run_case(ns, "tile260", ids, variants, 3)
    rng = np.random.default_rng(130)
    pops = np.repeat(np.arange(2), [66, 64])
    freqs = np.stack([np.where(pops == k, 0.4, 0.05) for k in range(2)] + [np.full(130, 0.1), np.full(130, 0.01)])
    ids, variants = make_variants(rng, 130, 2100, pop_freq=freqs)
    run_case(ns, "tile130", ids, variants, 4)


if __name__ == "__main__":
    main()
