import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def load_pkg(sub=None):
    """The package directory is 'spark-examples_amd' (hyphen) -> importlib."""
    name = "spark-examples_amd" + ("." + sub if sub else "")
    return importlib.import_module(name)


def load_oracle():
    """oracle/ is test infrastructure: imported by tests only."""
    odir = os.path.join(ROOT, "oracle")
    if odir not in sys.path:
        sys.path.insert(0, odir)
    return importlib.import_module("variants_pca_oracle")


def golden_cases():
    return sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz"))


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def oracle():
    return load_oracle()


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


def int_gram(x):
    """X^T X of a small-integer matrix as int64, through the fp64 BLAS (numpy's integer matmul is a scalar loop: 30 s at
    2504 x 3000).  Exact: every product and partial sum is an integer below 2^53, which is asserted."""
    xf = np.asarray(x, dtype=np.float64)
    assert xf.size == 0 or float(xf.max()) ** 2 * xf.shape[0] < 2.0 ** 53
    g = xf.T @ xf
    out = g.astype(np.int64)
    assert np.array_equal(out.astype(np.float64), g)
    return out


def align_sign(v, ref):
    """Eigenvectors are defined up to sign: flip columns of v to match ref."""
    v = np.array(v, dtype=np.float64, copy=True)
    for c in range(v.shape[1]):
        if np.dot(v[:, c], ref[:, c]) < 0:
            v[:, c] = -v[:, c]
    return v


def planted_callsets(rng, n, v, k=3, hi=0.5, lo=0.05):
    """Random carrier lists with planted population structure (clear spectral gaps)."""
    pops = np.sort(rng.integers(0, k, size=n))
    x = np.zeros((v, n), dtype=np.float32)
    for row in range(v):
        which = rng.integers(0, k + 1)
        p = np.where(pops == which, hi, lo) if which < k else np.full(n, rng.uniform(0.02, 0.4))
        x[row] = rng.random(n) < p
    return x


def write_golden_vcf(g, path, gz=False):
    """The variant records a golden fixture was generated from (`variants_json`: the dicts the reference's
    prepare_call_data consumed) as a VCF: one column per callset in callset-list order, a missing call as './.', a
    haploid / triploid genotype as is.  Lets the hosts' ingest be held to the reference's own outputs."""
    import gzip
    import json
    ids = [str(s) for s in g["callset_ids"]]
    col = dict((cid, i) for i, cid in enumerate(ids))
    variants = json.loads(str(g["variants_json"]))
    opener = gzip.open if gz else open
    with opener(path, "wt") as f:
        f.write("##fileformat=VCFv4.2\n")
        f.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" +
                "\t".join("S%04d" % i for i in range(len(ids))) + "\n")
        for k, var in enumerate(variants):
            cells = ["./."] * len(ids)
            for call in var.get("calls", []):
                cells[col[call["callSetId"]]] = ("|" if k % 2 else "/").join(str(a) for a in call["genotype"])
            f.write("chr17\t%d\t.\tA\tC\t.\tPASS\t.\tGT\t%s\n" % (41196312 + 7 * k, "\t".join(cells)))
    return len(ids)


def write_golden_plink(g, prefix, flip=False):
    """The same records as a PLINK 1 fileset (<prefix>.bed/.bim/.fam).  PLINK holds diploid biallelic genotypes only, so a
    genotype is mapped to the code with the same hasVariation (VariantsPca.scala:56-60): every allele > 0 -> homozygous
    non-reference, some allele > 0 -> heterozygous, no allele > 0 and one missing -> missing, else homozygous reference.
    A2 is the reference allele (flip=True: A1 is, i.e. the codes 00 and 11 trade places).  A callset without a call in a
    record is missing."""
    import json
    ids = [str(s) for s in g["callset_ids"]]
    col = dict((cid, i) for i, cid in enumerate(ids))
    variants = json.loads(str(g["variants_json"]))
    n = len(ids)
    hom_alt, het, missing, hom_ref = (3, 2, 1, 0) if flip else (0, 2, 1, 3)
    with open(prefix + ".fam", "w") as f:
        for i in range(n):
            f.write("FAM%d S%04d 0 0 0 -9\n" % (i, i))
    with open(prefix + ".bim", "w") as f:
        for k in range(len(variants)):
            f.write("17\trs%d\t0\t%d\tC\tA\n" % (k, 41196312 + 7 * k))
    bpv = (n + 3) // 4
    out = np.zeros((len(variants), bpv), dtype=np.uint8)
    for k, var in enumerate(variants):
        codes = np.full(bpv * 4, hom_ref, dtype=np.uint8)     # padding bits of the last byte: anything
        codes[:n] = missing
        for call in var.get("calls", []):
            gt = call["genotype"]
            if gt and all(a > 0 for a in gt):
                c = hom_alt
            elif any(a > 0 for a in gt):
                c = het
            elif any(a < 0 for a in gt) or not gt:
                c = missing
            else:
                c = hom_ref
            codes[col[call["callSetId"]]] = c
        quad = codes.reshape(bpv, 4)
        out[k] = quad[:, 0] | (quad[:, 1] << 2) | (quad[:, 2] << 4) | (quad[:, 3] << 6)
    with open(prefix + ".bed", "wb") as f:
        f.write(bytes([0x6c, 0x1b, 0x01]))
        f.write(out.tobytes())
    return n
