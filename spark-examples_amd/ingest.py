"""Local data sources for the driver (host-side ETL; not on the GPU hot path).

The reference streamed variants from the Google Genomics API (VariantsRDD.scala:205-235) and listed
callsets over REST (VariantsCommon.scala:38-50); that service is gone.  These loaders reproduce the
semantics that matter downstream: callset index = position in the callset list
(VariantsCommon.scala:44-45), hasVariation = any allele index > 0 (VariantsPca.scala:56-60),
variants without a varying call dropped (VariantsPca.scala:164-167), `dataset` = callset id up to
the first '-' (VariantsPca.scala:235).
"""
import gzip
import os
import re

import numpy as np

from . import synth

# contig normalisation of the reference (VariantsRDD.scala:103-110): only [a-z]*[0-9]* survives,
# which drops X / Y / MT; the numeric part is the contig key.
_CONTIG_RE = re.compile(r"^([a-z]*)?([0-9]+)$")


def normalize_contig(name):
    m = _CONTIG_RE.match(name.lower())
    return m.group(2) if m else None


# A .bim names the sex and mitochondrial chromosomes by number; the reference's contig rule drops X / Y / MT
# (VariantsRDD.scala:103-110), so the numeric codes are mapped back to their names before that rule is applied and the same
# cohort gives the same S as a VCF and as a PLINK fileset.
_PLINK_CHROM = {"23": "X", "24": "Y", "25": "XY", "26": "MT", "0": "unplaced"}


def plink_contig(chrom):
    return normalize_contig(_PLINK_CHROM.get(chrom, chrom))


def parse_references(refs):
    """'chr17:41196311:41277499' (optionally comma separated) -> [(contig, start, end)]"""
    out = []
    for item in refs or []:
        for tup in item.split(","):
            if not tup:
                continue
            contig, start, end = tup.split(":")
            out.append((normalize_contig(contig) or contig, int(start), int(end)))
    return out


def set_id_of(path, ordinal=0, used=None):
    """Set id of a VCF = its file name up to the first '.', '-' replaced ('-' separates the dataset name from the
    callset number in an id, VariantsPca.scala:235).  A stem already taken by an earlier variant set of the same run
    (a/cohort.chr17.vcf + b/cohort.chr17.vcf) gets '_<ordinal>' appended -- the rule of the compiled host."""
    stem = os.path.basename(path).split(".")[0].replace("-", "_")
    if used is not None:
        if stem in used:
            stem = "%s_%d" % (stem, ordinal)
        used.add(stem)
    return stem


def load_npz(path):
    z = np.load(path, allow_pickle=False)
    ids = [str(s) for s in z["callset_ids"]]
    names = [str(s) for s in z["callset_names"]] if "callset_names" in z.files else ids
    indexes = dict((cid, i) for i, cid in enumerate(ids))
    name_map = dict(zip(ids, names))
    return indexes, name_map, [("csr", z["sample_idx"].astype(np.int32), z["row_offsets"].astype(np.int64))]


def _carriers_of_cells(cells, gti, n_samples):
    """Sample columns of one VCF data line (bytes after the 9th tab) -> indices of the calls with variation.
    hasVariation = some allele index > 0 (VariantsPca.scala:56-60) = a digit 1-9 inside the GT sub-field (sub-field
    number gti of the ':'-separated column; '.', '0' alleles and phasing marks do not count).  numpy over the bytes:
    the column of a byte is the count of tabs before it, its sub-field the count of ':' since that tab."""
    a = np.frombuffer(cells, dtype=np.uint8)
    if a.size == 0:
        return np.zeros(0, dtype=np.int32)
    tabs = a == 9
    cell = np.cumsum(tabs)                      # column of every byte (a tab itself counts to the next: harmless)
    digit = (a >= 49) & (a <= 57)
    if gti == 0:
        colons = np.cumsum(a == 58)
        base = np.concatenate(([0], colons[tabs]))          # ':' seen before each column starts
        hit = digit & (colons == base[cell])
    else:
        colons = np.cumsum(a == 58)
        base = np.concatenate(([0], colons[tabs]))
        hit = digit & (colons - base[cell] == gti)
    cols = np.unique(cell[hit])
    return cols[cols < n_samples].astype(np.int32)


def load_vcf(path, references=None):
    """Minimal VCF reader: GT only.  Returns CSR carrier lists directly (one numpy pass per line over the sample
    columns; ~40x the per-cell Python loop it replaced, same carriers as the compiled host, tests/test_host.py)."""
    regions = parse_references(references)
    opener = gzip.open if path.endswith(".gz") else open
    set_id = set_id_of(path)
    samples = None
    idx_chunks, offs = [], [0]
    with opener(path, "rb") as f:
        for raw in f:
            if raw.startswith(b"##"):
                continue
            line = raw.rstrip(b"\n")
            if line.startswith(b"#CHROM"):
                samples = line.decode().split("\t")[9:]
                continue
            if samples is None:
                raise ValueError("VCF header line (#CHROM) missing")
            head = line.split(b"\t", 9)
            if len(head) < 10:
                continue
            contig = normalize_contig(head[0].decode())
            if contig is None:
                continue  # the reference drops contigs such as X, Y, MT (VariantsRDD.scala:132-136)
            start = int(head[1]) - 1
            if regions and not any(c == contig and s <= start < e for (c, s, e) in regions):
                continue
            fmt = head[8].split(b":")
            if b"GT" not in fmt:
                continue
            carriers = _carriers_of_cells(head[9], fmt.index(b"GT"), len(samples))
            if carriers.size:
                idx_chunks.append(carriers)
                offs.append(offs[-1] + int(carriers.size))
    if samples is None:
        raise ValueError("no #CHROM header in %s" % path)
    ids = ["%s-%d" % (set_id, i) for i in range(len(samples))]
    idx = np.concatenate(idx_chunks) if idx_chunks else np.zeros(0, dtype=np.int32)
    indexes = dict((cid, i) for i, cid in enumerate(ids))
    return indexes, dict(zip(ids, samples)), [("csr", idx, np.asarray(offs, dtype=np.int64))]


def load_plink(path, references=None, ref_allele="a2", chunk_variants=4096, as_bits=False, as_bed=False):
    """PLINK 1 binary fileset (<prefix>.bed / .bim / .fam; `path` is the prefix or any of the three files).  The .bed is
    variant-major, two bits per genotype, four samples to a byte (sample s in bits 2 (s % 4) of byte s // 4):
    00 homozygous A1, 01 missing, 10 heterozygous, 11 homozygous A2.  hasVariation (VariantsPca.scala:56-60: some allele
    index > 0) = the genotype carries a non-reference allele; the reference allele is A2 (what `plink --keep-allele-order`
    and `plink2 --make-bed` write for a VCF's REF; ref_allele="a1" for filesets written the other way round), so codes
    00 and 10 count and a missing call does not (its allele indices are -1).  Returns what load_vcf returns: CSR carrier
    lists of the variants that have a carrier, callset index = row of the .fam, name = its IID.  The contig rule and the
    --references filter are the VCF reader's (0-based start = bp - 1).
    as_bits=True returns [("bits", uint32 [variants][ceil(N / 32)])] instead: the carrier bitsets pcoa_accumulate_bits
    takes (sample i -> bit i & 31 of word i >> 5), one table look-up per .bed byte and no CSR in between -- 1 bit per
    genotype instead of 4 bytes per carrier; rows without a carrier stay (they add nothing to S).
    as_bed=True (r05) decodes nothing at all: [("bed", memory-mapped uint8 [variants][ceil(N / 4)] rows as they lie in the
    file, keep mask over the .bim lines, ref_is_a1)] -- what pcoa_accumulate_plink_bed takes block by block (the device
    decodes; the numpy decode above runs at ~0.2 M variants/s, a memory-mapped block copy at GB/s)."""
    prefix = path[:-4] if path[-4:] in (".bed", ".bim", ".fam") else path
    regions = parse_references(references)
    set_id = set_id_of(prefix + ".bed")
    with open(prefix + ".fam") as f:
        names = [ln.split()[1] for ln in f if ln.strip()]
    n = len(names)
    if n == 0:
        raise ValueError("no samples in %s.fam" % prefix)
    keep = []
    with open(prefix + ".bim") as f:
        for ln in f:
            if not ln.strip():
                continue
            t = ln.split()
            contig = plink_contig(t[0])
            ok = contig is not None
            if ok and regions:
                start = int(t[3]) - 1
                ok = any(c == contig and s0 <= start < e for (c, s0, e) in regions)
            keep.append(ok)
    keep = np.asarray(keep, dtype=bool)
    bpv = (n + 3) // 4                                          # bytes per variant
    raw = np.memmap(prefix + ".bed", dtype=np.uint8, mode="r")     # a cohort's .bed can be tens of GB: read in chunks
    if raw.size < 3 or raw[0] != 0x6c or raw[1] != 0x1b:
        raise ValueError("%s.bed: not a PLINK 1 binary file" % prefix)
    if raw[2] != 1:
        raise ValueError("%s.bed is sample-major; only the variant-major layout (plink >= 1.07 default) is read" % prefix)
    if raw.size - 3 != keep.size * bpv:
        raise ValueError("%s.bed holds %d bytes of genotypes, %d variants x %d samples need %d"
                         % (prefix, raw.size - 3, keep.size, n, keep.size * bpv))
    geno = raw[3:].reshape(keep.size, bpv)
    ids = ["%s-%d" % (set_id, i) for i in range(n)]
    indexes = dict((cid, i) for i, cid in enumerate(ids))
    if as_bed:
        return indexes, dict(zip(ids, names)), [("bed", geno, keep, ref_allele != "a2")]
    if as_bits:
        # a genotype varies iff the LOW bit of its code is 0 (A2 = reference: codes 00, 10) / the HIGH bit is 1 (A1 =
        # reference: codes 10, 11): one nibble of carrier bits per .bed byte, eight nibbles to a word
        b = np.arange(256, dtype=np.uint32)
        if ref_allele == "a2":
            lut = sum((1 - ((b >> (2 * q)) & 1)) << q for q in range(4))
        else:
            lut = sum(((b >> (2 * q + 1)) & 1) << q for q in range(4))
        lut = lut.astype(np.uint32)
        words = (n + 31) // 32
        bits = np.zeros((int(keep.sum()), words), dtype=np.uint32)
        done = 0
        for v0 in range(0, keep.size, chunk_variants):
            k = keep[v0:v0 + chunk_variants]
            if not k.any():
                continue
            g = np.asarray(geno[v0:v0 + chunk_variants])[k]
            nib = np.zeros((g.shape[0], words * 8), dtype=np.uint32)
            nib[:, :bpv] = lut[g]
            nib = nib.reshape(g.shape[0], words, 8)
            out = bits[done:done + g.shape[0]]
            for q in range(8):
                out |= nib[:, :, q] << np.uint32(4 * q)
            done += g.shape[0]
        if n % 32:
            bits[:, -1] &= np.uint32((1 << (n % 32)) - 1)        # the codes behind the last sample are padding
        return indexes, dict(zip(ids, names)), [("bits", bits)]
    varies = (0, 2) if ref_allele == "a2" else (3, 2)           # codes that carry a non-reference allele
    shifts = np.array([0, 2, 4, 6], dtype=np.uint8)
    idx_chunks, counts = [], []
    for v0 in range(0, keep.size, chunk_variants):
        k = keep[v0:v0 + chunk_variants]
        if not k.any():
            continue
        g = np.asarray(geno[v0:v0 + chunk_variants])[k]
        codes = ((g[:, :, None] >> shifts[None, None, :]) & 3).reshape(g.shape[0], -1)[:, :n]
        has = (codes == varies[0]) | (codes == varies[1])
        cnt = has.sum(axis=1)
        rows, cols = np.nonzero(has)
        idx_chunks.append(cols.astype(np.int32))
        counts.append(cnt[cnt > 0])                             # variants without a carrier are dropped (:164-167)
    idx = np.concatenate(idx_chunks) if idx_chunks else np.zeros(0, dtype=np.int32)
    cnts = np.concatenate(counts) if counts else np.zeros(0, dtype=np.int64)
    offs = np.concatenate(([0], np.cumsum(cnts))).astype(np.int64)
    return indexes, dict(zip(ids, names)), [("csr", idx, offs)]


def load_vcf_records(path, references=None, set_id=None):
    """VCF -> variant records shaped like the reference's Variant case class (VariantsRDD.scala:46-54):
    contig, start, end, referenceBases, alternateBases, info, calls[{callSetId, genotype}].
    Returns (callset ids in file order, id -> name, [variant dict])."""
    regions = parse_references(references)
    opener = gzip.open if path.endswith(".gz") else open
    if set_id is None:
        set_id = set_id_of(path)
    samples, ids, variants = None, [], []
    with opener(path, "rt") as f:
        for line in f:
            if line.startswith("##"):
                continue
            if line.startswith("#CHROM"):
                samples = line.rstrip("\n").split("\t")[9:]
                ids = ["%s-%d" % (set_id, i) for i in range(len(samples))]
                continue
            if samples is None:
                raise ValueError("VCF header line (#CHROM) missing")
            rec = line.rstrip("\n").split("\t")
            if len(rec) < 10:
                continue
            contig = normalize_contig(rec[0])
            if contig is None:
                continue
            start = int(rec[1]) - 1
            if regions and not any(c == contig and s <= start < e for (c, s, e) in regions):
                continue
            fmt = rec[8].split(":")
            if "GT" not in fmt:
                continue
            gti = fmt.index("GT")
            info = {}
            for item in rec[7].split(";"):
                if "=" in item:
                    k, val = item.split("=", 1)
                    info[k] = val.split(",")
            calls = []
            for i, cell in enumerate(rec[9:]):
                parts = cell.split(":")
                gt = parts[gti] if gti < len(parts) else "."
                genotype = [(-1 if a in (".", "") else int(a)) for a in re.split(r"[/|]", gt)]
                calls.append({"callSetId": ids[i], "genotype": genotype})
            variants.append({"contig": contig, "start": start, "end": start + len(rec[3]), "referenceBases": rec[3],
                             "alternateBases": [a for a in rec[4].split(",") if a != "."], "info": info,
                             "calls": calls})
    if samples is None:
        raise ValueError("no #CHROM header in %s" % path)
    return ids, dict(zip(ids, samples)), variants


def synthetic_dataset(spec):
    """'V,N,seed' -> Balding-Nichols synthetic carriers as CSR (host generated; small V only)."""
    v, n, seed = [int(t) for t in spec.split(",")]
    offsets = synth.pop_offsets(n)
    thr = synth.thresholds(seed, 0, v, n_pops=len(offsets) - 1)
    x = synth.genotypes(seed, 0, thr, offsets, dtype=np.uint8)
    keep = x.any(axis=1)
    x = x[keep]
    counts = x.sum(axis=1).astype(np.int64)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    idx = np.nonzero(x)[1].astype(np.int32)
    pops = ["AFR", "AMR", "EAS", "EUR", "SAS"]
    ids, names = [], {}
    for i in range(n):
        p = int(np.searchsorted(offsets, i, side="right") - 1)
        cid = "%s-%d" % (pops[p % len(pops)], i)
        ids.append(cid)
        names[cid] = "S%06d" % i
    indexes = dict((cid, i) for i, cid in enumerate(ids))
    return indexes, names, [("csr", idx, offs)]


def pack_bits(x, pad_words=0):
    """Dense 0/1 [variants][samples] -> carrier bitsets uint32 [variants][ceil(N/32) + pad_words] in the layout of
    pcoa_accumulate_bits: sample i = bit (i & 31) of word i >> 5 (little-endian bit order)."""
    x = np.asarray(x)
    v, n = x.shape
    words = (n + 31) // 32
    b = np.packbits(x.astype(bool), axis=1, bitorder="little")          # [v][ceil(n/8)] bytes
    out = np.zeros((v, (words + pad_words) * 4), dtype=np.uint8)
    out[:, : b.shape[1]] = b
    return out.view("<u4")

