"""CPU tests of the host side: C-ABI surface, reference-interface mirror, synthetic generator,
sharding logic.  No compute call reaches a GPU here."""
import io
import json
import os
import re
import sys

import numpy as np
import pytest

from conftest import ROOT, golden_cases, load_golden, load_pkg


def test_library_loads_and_exports_every_declared_symbol():
    lib_mod = load_pkg("_lib")
    lib = lib_mod.load()
    header = open(os.path.join(ROOT, "include", "pcoa.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(pcoa_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), "libpcoa_hip.so does not export %s" % name
    assert set(lib_mod.EXPORTED_SYMBOLS) == declared
    assert b"gfx950" in lib.pcoa_version()


def test_header_is_plain_c_and_a_c_client_links_against_the_library(tmp_path):
    """include/pcoa.h is the drop-in boundary: it must compile as strict C99 (no C++ types in the signatures) and a
    plain C program must link against libpcoa_hip.so -- what a cgo / JNI / FFI binding does.  Without a GPU the
    client sees PCOA_ERR_NO_DEVICE and a message, never a crash or a CPU fallback."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("gcc not available")
    src = tmp_path / "client.c"
    src.write_text(
        '#include <stdio.h>\n#include "pcoa.h"\n'
        "int main(void) {\n"
        "  pcoa_ctx* ctx = NULL;\n"
        "  pcoa_timings t; pcoa_synth_params sp; (void)t; (void)sp;\n"
        '  printf("%s\\n", pcoa_version());\n'
        "  int rc = pcoa_create(&ctx, 8, 0, PCOA_FLAG_DEFAULT);\n"
        '  printf("rc=%d %s\\n", rc, pcoa_last_error(NULL));\n'
        "  if (rc == PCOA_OK) { pcoa_destroy(ctx); return 0; }\n"
        "  return (rc == PCOA_ERR_NO_DEVICE && ctx == NULL) ? 0 : 1;\n"
        "}\n")
    libdir = os.path.join(ROOT, "spark-examples_amd")
    exe = str(tmp_path / "client")
    res = subprocess.run([gcc, "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                          str(src), "-o", exe, "-L", libdir, "-lpcoa_hip", "-Wl,-rpath," + libdir,
                          "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert res.returncode == 0, res.stdout
    run = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert run.returncode == 0, run.stdout
    assert "gfx950" in run.stdout


def test_no_cpu_fallback_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    pkg = load_pkg()
    with pytest.raises(pkg.PcoaError) as ei:
        pkg.PcoaEngine(8)
    assert ei.value.code == load_pkg("_lib").PCOA_ERR_NO_DEVICE
    assert "no CPU fallback" in str(ei.value)


def test_product_never_imports_the_oracle():
    pdir = os.path.join(ROOT, "spark-examples_amd")
    for dirpath, _, files in os.walk(pdir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp", "Makefile")):
                src = open(os.path.join(dirpath, f)).read()
                assert "variants_pca_oracle" not in src and "pcoa_oracle" not in src, f
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f


@pytest.mark.parametrize("name", golden_cases())
def test_prepare_call_data_matches_reference_python(name):
    vp = load_pkg("variants_pca")
    g = load_golden(name)
    variants = json.loads(str(g["variants_json"]))
    ids = [str(s) for s in g["callset_ids"]]
    id_to_index = dict((c, i) for i, c in enumerate(ids))
    calls = vp.prepare_call_data(variants, id_to_index)
    offs = g["row_offsets"]
    assert len(calls) == len(offs) - 1
    for v, c in enumerate(calls):
        assert list(c) == list(g["sample_idx"][offs[v]:offs[v + 1]])


def test_extract_call_info_follows_the_scala_driver():
    vp = load_pkg("variants_pca")
    mapping = {"a-0": 0, "a-1": 1, "a-2": 2}
    variant = {"calls": [{"callSetId": "a-0", "genotype": [0, 1]}, {"callSetId": "a-1", "genotype": [-1, -1]},
                         {"callSetId": "a-2", "genotype": [0, 0]}]}
    assert vp.extract_call_info(variant, mapping) == [(True, 0), (False, 1), (False, 2)]
    assert vp.extract_call_info({}, mapping) == []
    with pytest.raises(KeyError):  # mapping(call.callsetId) throws in the reference
        vp.extract_call_info({"calls": [{"callSetId": "zzz", "genotype": [1]}]}, mapping)


def test_java_double_to_string():
    f = load_pkg("variants_pca").java_double_to_string
    assert f(0.0286308791579312) == "0.0286308791579312"      # README.md:108-120 sample rows
    assert f(-0.008456233951873527) == "-0.008456233951873527"
    assert f(1.234e-4) == "1.234E-4" and f(1.0) == "1.0" and f(12345678.9) == "1.23456789E7"
    assert f(0.001) == "0.001" and f(9.999e-4) == "9.999E-4" and f(0.0) == "0.0" and f(100.0) == "100.0"
    g = load_pkg("variants_pca").java_float_to_string
    assert g(0.1) == "0.1" and g(0.05) == "0.05" and g(1e-4) == "1.0E-4" and g(0.25) == "0.25" and g(3) == "3.0"


def test_pcaconf_flags_and_defaults():
    vp = load_pkg("variants_pca")
    c = vp.PcaConf([])
    assert c.numPc == 2 and c.num_reduce_partitions == 10 and c.bases_per_partition == 1000000
    assert c.references == ["chr17:41196311:41277499"] and c.variant_set_id == ["3049512673186936334"]
    c = vp.PcaConf(["--num-pc", "3", "--output-path", "/tmp/x", "--min-allele-frequency", "0.05",
                    "--spark-master", "local[4]", "--all-references", "--debug-datasets",
                    "--variant-set-id", "a", "b", "--input-path", "f.npz", "--client-secrets", "s.json"])
    assert c.numPc == 3 and c.outputPath == "/tmp/x" and abs(c.minAlleleFrequency - 0.05) < 1e-12
    assert c.variant_set_id == ["a", "b"] and c.all_references and c.debug_datasets


def test_emit_result_format(tmp_path):
    vp = load_pkg("variants_pca")
    conf = vp.PcaConf(["--output-path", str(tmp_path / "out")])
    drv = vp.VariantsPcaDriver.__new__(vp.VariantsPcaDriver)
    drv.conf, drv.indexes, drv.engine = conf, {"ds1-7": 0, "ds2-3": 1}, None
    drv.names = {"ds1-7": "NA20811", "ds2-3": "HG00096"}
    buf = io.StringIO()
    drv.emitResult([("ds1-7", 0.0286308791579312, -0.008456233951873527), ("ds2-3", -1e-4, 0.5)], out=buf)
    assert buf.getvalue() == ("HG00096\tds2\t-1.0E-4\t0.5\n"
                              "NA20811\tds1\t0.0286308791579312\t-0.008456233951873527\n")
    saved = open(str(tmp_path / "out") + "-pca.tsv").read().splitlines()
    assert saved[0] == "HG00096\t-1.0E-4\t0.5\tds2"


def test_philox_known_answers_and_generator_shard_invariance():
    synth = load_pkg("synth")
    r = synth.philox4x32_10(0, 0, 0, 0, 0, 0)
    assert [int(x) for x in r] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    r = synth.philox4x32_10(*([0xffffffff] * 6))
    assert [int(x) for x in r] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    r = synth.philox4x32_10(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0)
    assert [int(x) for x in r] == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    offs = synth.pop_offsets(2504)
    assert list(np.diff(offs)) == [661, 347, 504, 503, 489]
    whole_t = synth.thresholds(1002, 0, 300)
    part_t = synth.thresholds(1002, 100, 50)
    assert np.array_equal(whole_t[100:150], part_t)
    small = synth.pop_offsets(50)
    whole = synth.genotypes(1002, 0, synth.thresholds(1002, 0, 300), small)
    part = synth.genotypes(1002, 100, part_t, small)
    assert np.array_equal(whole[100:150], part)
    assert 0.02 < whole.mean() < 0.4


def test_shard_range_partitions_exactly():
    dist = load_pkg("dist")
    for v in (0, 1, 7, 1000, 1000003):
        for w in (1, 2, 3, 8):
            ranges = [dist.shard_range(r, w, v) for r in range(w)]
            assert ranges[0][0] == 0 and ranges[-1][1] == v
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        dist.shard_range(2, 2, 10)


def test_vcf_ingest(tmp_path):
    ingest = load_pkg("ingest")
    vcf = tmp_path / "brca1.vcf"
    vcf.write_text(
        "##fileformat=VCFv4.2\n"
        "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tNA1\tNA2\tNA3\n"
        "17\t41196312\t.\tA\tG\t.\tPASS\tAF=0.5\tGT:DP\t0|1:3\t0/0:4\t./.:0\n"
        "chr17\t41196400\t.\tC\tT,G\t.\tPASS\t.\tGT\t0|0\t2|0\t1\n"
        "17\t50000000\t.\tC\tT\t.\tPASS\t.\tGT\t1|1\t1|1\t1|1\n"      # outside the region
        "X\t100\t.\tC\tT\t.\tPASS\t.\tGT\t1|1\t1|1\t1|1\n"              # contig dropped by the reference
        "17\t41196500\t.\tC\tT\t.\tPASS\t.\tGT\t0|0\t0|0\t.\n")        # nobody varies: row dropped
    indexes, names, data = ingest.load_vcf(str(vcf), ["chr17:41196311:41277499"])
    assert [names[k] for k in sorted(indexes, key=indexes.get)] == ["NA1", "NA2", "NA3"]
    _, idx, offs = data[0]
    assert list(offs) == [0, 1, 3] and list(idx) == [0, 1, 2]
    assert all(k.split("-")[0] == "brca1" for k in indexes)


def test_murmur3_128_matches_guava_known_answers():
    vp = load_pkg("variants_pca")
    assert vp.murmur3_128_hex(b"") == "00000000000000000000000000000000"
    # Guava: Hashing.murmur3_128().hashString("hello", UTF_8).toString()
    assert vp.murmur3_128_hex(b"hello") == "029bbd41b3a7d8cb191dae486a901e5b"
    # MurmurHash3_x64_128 reference vector ("The quick brown fox jumps over the lazy dog", seed 0)
    assert vp.murmur3_128_hex(b"The quick brown fox jumps over the lazy dog") == "6c1b07bc7bbc4be347939ac4a93c437a"
    k1 = vp.get_variant_key({"contig": "17", "start": 41196311, "end": 41196312, "referenceBases": "A",
                             "alternateBases": ["G"]})
    k2 = vp.get_variant_key({"contig": "17", "start": 41196311, "end": 41196312, "referenceBases": "A",
                             "alternateBases": ["G", "T"]})
    assert len(k1) == 32 and k1 != k2


def _variant(contig, start, ref, alts, calls, af=None):
    v = {"contig": contig, "start": start, "end": start + len(ref), "referenceBases": ref, "alternateBases": alts,
         "calls": [{"callSetId": c, "genotype": g} for c, g in calls]}
    if af is not None:
        v["info"] = {"AF": [str(af)]}
    return v


def test_join_and_merge_follow_the_reference_semantics():
    vp = load_pkg("variants_pca")
    idx = {"a-0": 0, "a-1": 1, "b-0": 2, "c-0": 3}
    a = [_variant("17", 10, "A", ["G"], [("a-0", [0, 1]), ("a-1", [0, 0])]),
         _variant("17", 20, "C", ["T"], [("a-0", [1, 1]), ("a-1", [0, 1])]),
         _variant("17", 30, "G", ["A"], [("a-0", [0, 0]), ("a-1", [0, 0])])]
    b = [_variant("17", 10, "A", ["G"], [("b-0", [1, 0])]),
         _variant("17", 20, "C", ["G"], [("b-0", [1, 1])]),          # different ALT: no match
         _variant("17", 30, "G", ["A"], [("b-0", [0, 0])])]          # nobody varies: dropped at :164-166
    c = [_variant("17", 10, "A", ["G"], [("c-0", [0, 1])]), _variant("17", 30, "G", ["A"], [("c-0", [1, 1])])]
    joined = vp.join_datasets([a, b], idx)
    assert sorted(map(sorted, joined)) == sorted(map(sorted, [[(True, 0), (False, 1), (True, 2)],
                                                               [(False, 0), (False, 1), (False, 2)]]))
    drv = vp.VariantsPcaDriver.__new__(vp.VariantsPcaDriver)
    drv.conf, drv.indexes = vp.PcaConf([]), idx
    assert sorted(map(sorted, drv.getCallsRdd([a, b]))) == [[0, 2]]
    merged = drv.getCallsRdd([a, b, c])                                # keys present in all three sets
    assert sorted(map(sorted, merged)) == [[0, 2, 3], [3]]
    # AF filter (VariantsPca.scala:96-108): variants without AF are dropped, comparison in float
    drv.conf = vp.PcaConf(["--min-allele-frequency", "0.05"])
    data = [_variant("17", 1, "A", ["G"], [], af=0.05), _variant("17", 2, "A", ["G"], [], af=0.01),
            _variant("17", 3, "A", ["G"], [])]
    assert [v["start"] for v in drv.filterDataset(data)] == [1]


def test_similarity_matrix_stream_form_keeps_only_nonzero_keys():
    """getSimilarityMatrixStream (VariantsPca.scala:262-279): pairs c1 <= c2 emitted per variant, reduced, strict upper
    entries mirrored -- the same counts as getSimilarityMatrix, keys with a zero count absent."""
    vp = load_pkg("variants_pca")
    callsets = [[0, 1], [0, 1, 2], [4], [0, 0, 1]]         # sample 3 never appears; the last row repeats a callset
    n = 5
    full = np.zeros((n, n), dtype=np.int64)
    stream = {}
    for c in callsets:
        for c1 in c:
            for c2 in c:
                full[c1, c2] += 1                           # :187
                if c1 <= c2:
                    stream[(c1, c2)] = stream.get((c1, c2), 0) + 1   # :267-268
    mirrored = dict(stream)
    for (i, j), v in stream.items():
        if i < j:
            mirrored[(j, i)] = v                            # :272-278
    got = dict(vp.similarity_entries_nonzero(full))
    assert got == mirrored
    assert all(3 not in k for k in got) and (0, 0) in got and got[(0, 0)] == 6


def test_pack_bits_layout_is_the_one_pcoa_accumulate_bits_documents():
    """include/pcoa.h: sample i of variant v is bit (i & 31) of word bits[v * ld_words + (i >> 5)]."""
    ingest = load_pkg("ingest")
    rng = np.random.default_rng(9)
    for n in (1, 31, 32, 33, 70, 2504):
        x = (rng.random((7, n)) < 0.4).astype(np.uint8)
        for pad in (0, 3):
            b = ingest.pack_bits(x, pad_words=pad)
            assert b.dtype == np.dtype("<u4") and b.shape == (7, (n + 31) // 32 + pad)
            for v in range(7):
                for i in range(n):
                    assert ((int(b[v, i >> 5]) >> (i & 31)) & 1) == x[v, i]
            assert not b[:, (n + 31) // 32:].any()
            if n % 32:
                assert not (b[:, (n - 1) >> 5] >> np.uint32(n % 32)).any()   # tail bits of the last word are zero
    # a carrier bitset is the CSR row of pcoa_accumulate_calls as a set
    callsets = [[0, 5, 33], [], [69]]
    dense = np.zeros((3, 70), dtype=np.uint8)
    for v, c in enumerate(callsets):
        dense[v, c] = 1
    b = ingest.pack_bits(dense)
    assert b[0, 0] == (1 | (1 << 5)) and b[0, 1] == (1 << 1) and not b[1].any() and b[2, 2] == (1 << 5)


def test_compiled_host_ingest_matches_python_ingest(tmp_path):
    """variants_pca_driver --parse-only (no GPU): the zero-allocation, multi-threaded VCF reader of the C++ host yields
    the same carrier lists (getCallsRdd, VariantsPca.scala:153-168) as the Python mirror, on awkward records."""
    import gzip
    import subprocess
    ingest = load_pkg("ingest")
    drv = os.path.join(ROOT, "spark-examples_amd", "variants_pca_driver")
    if not os.path.exists(drv):
        pytest.skip("compiled host not built")
    rng = np.random.default_rng(21)
    n = 37
    names = ["S%02d" % i for i in range(n)]
    gts = ["0|0", "0|1", "1|0", "1|1", "./.", ".", "0/2", "10|0", "0", "1", "0|0|0", "0/0/3", ".|1"]
    lines = ["##fileformat=VCFv4.2", "##contig=<ID=17>",
             "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names)]
    for r in range(300):
        fmt = ["GT", "GT:DP", "DP:GT", "DP:GT:GQ", "DP"][r % 5]
        cells = []
        for i in range(n):
            gt = gts[int(rng.integers(len(gts)))] if rng.random() < 0.4 else "0|0"
            cell = {"GT": gt, "GT:DP": gt + ":17", "DP:GT": "9:" + gt, "DP:GT:GQ": "31:" + gt + ":99", "DP": "5"}[fmt]
            if fmt == "DP:GT:GQ" and i % 7 == 0:
                cell = "31"           # trailing sub-fields dropped: GT missing for this call
            cells.append(cell)
        if r % 50 == 49:
            cells = cells[:n - 5]     # a short record
        chrom = ["17", "chr17", "X", "17"][r % 4]
        lines.append("\t".join([chrom, str(41196312 + 13 * r), ".", "A", "G,T" if r % 3 == 0 else "G", "50", "PASS",
                                "AF=0.1;DP=4", fmt] + cells))
    text = "\n".join(lines) + "\n"
    plain = str(tmp_path / "awk-ward.vcf")
    open(plain, "w").write(text)
    crlf = str(tmp_path / "crlf.vcf")
    open(crlf, "w").write(text.replace("\n", "\r\n"))
    gz = str(tmp_path / "zipped.vcf.gz")
    with gzip.open(gz, "wt") as f:
        f.write(text)
    for path in (plain, crlf, gz):
        for threads in ("1", "3"):
            out = str(tmp_path / "o")
            res = subprocess.run([drv, "--input-path", path, "--all-references", "--parse-only", "--ingest-threads", threads,
                                  "--output-path", out], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                 universal_newlines=True)
            assert res.returncode == 0, res.stdout
            assert "Matrix size: %d." % n in res.stdout
            got = [[int(t) for t in l.split()] for l in open(out + "-carriers.txt").read().splitlines()]
            ref_path = plain if path == crlf else path   # the Python reader keeps the \r of the last column
            _, _, parts = ingest.load_vcf(ref_path, None)
            _, idx, offs = parts[0]
            want = [idx[offs[k]:offs[k + 1]].tolist() for k in range(len(offs) - 1)]
            assert got == want, (path, threads)
            assert len(want) > 100


def _driver_exe():
    drv = os.path.join(ROOT, "spark-examples_amd", "variants_pca_driver")
    if not os.path.exists(drv):
        pytest.skip("compiled host not built")
    return drv


def _parse_only_rows(drv, paths, out, extra=()):
    import subprocess
    res = subprocess.run([drv, "--input-path"] + list(paths) + ["--parse-only", "--output-path", out] + list(extra),
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert res.returncode == 0, res.stdout
    return res.stdout, [[int(t) for t in l.split()] for l in open(out + "-carriers.txt").read().splitlines()]


@pytest.mark.parametrize("name", golden_cases())
def test_vcf_ingest_of_both_hosts_reproduces_the_reference_carrier_rows(name, tmp_path):
    """SURVEY 8(f) rank 1 pinned to the reference: the variant records each golden fixture was generated from are
    written as a VCF; the C++ host's and the Python mirror's ingest + getCallsRdd must return exactly the rows the
    reference's own prepare_call_data produced from those records (tests/golden/make_golden.py)."""
    from conftest import write_golden_vcf
    g = load_golden(name)
    offs = g["row_offsets"]
    want = [g["sample_idx"][offs[k]:offs[k + 1]].tolist() for k in range(len(offs) - 1)]
    n = int(g["n_samples"])
    ingest = load_pkg("ingest")
    for gz in (False, True):
        path = str(tmp_path / ("golden.vcf.gz" if gz else "golden.vcf"))
        assert write_golden_vcf(g, path, gz=gz) == n
        _, _, parts = ingest.load_vcf(path, ["chr17:41196311:41277499"])
        _, idx, o = parts[0]
        assert [idx[o[k]:o[k + 1]].tolist() for k in range(len(o) - 1)] == want
        stdout, got = _parse_only_rows(_driver_exe(), [path], str(tmp_path / "o"))
        assert "Matrix size: %d." % n in stdout
        assert got == want


@pytest.mark.parametrize("name", golden_cases())
def test_plink_fileset_ingest_reproduces_the_reference_carrier_rows(name, tmp_path):
    """SURVEY 8(f) rank 1, the cohort format PCA users actually hold: the records of each golden fixture as a PLINK 1
    fileset (.bed two bits per genotype, all four codes present); ingest.load_plink must return the carrier rows the
    reference's own prepare_call_data produced, whichever allele the fileset calls the reference."""
    from conftest import write_golden_plink
    g = load_golden(name)
    offs = g["row_offsets"]
    want = [g["sample_idx"][offs[k]:offs[k + 1]].tolist() for k in range(len(offs) - 1)]
    n = int(g["n_samples"])
    ingest = load_pkg("ingest")
    for flip in (False, True):
        prefix = str(tmp_path / ("flip" if flip else "plain"))
        assert write_golden_plink(g, prefix, flip=flip) == n
        indexes, names, parts = ingest.load_plink(prefix + ".bed", ["chr17:41196311:41277499"],
                                                  ref_allele="a1" if flip else "a2", chunk_variants=7)
        kind, idx, o = parts[0]
        assert kind == "csr" and idx.dtype == np.int32 and o.dtype == np.int64
        assert [idx[o[k]:o[k + 1]].tolist() for k in range(len(o) - 1)] == want
        assert len(indexes) == n and sorted(indexes.values()) == list(range(n))
        stem = "flip" if flip else "plain"
        assert indexes["%s-3" % stem] == 3 and names["%s-3" % stem] == "S0003"
        # the bitset form (what the driver feeds to pcoa_accumulate_bits): the same carriers, every variant kept
        _, _, bparts = ingest.load_plink(prefix + ".bed", ["chr17:41196311:41277499"],
                                         ref_allele="a1" if flip else "a2", as_bits=True, chunk_variants=5)
        assert bparts[0][0] == "bits" and bparts[0][1].dtype == np.uint32 and bparts[0][1].shape[1] == (n + 31) // 32
        rows = [np.nonzero((np.asarray(r)[:, None] >> np.arange(32, dtype=np.uint32)[None, :]).reshape(-1)[:n] & 1)[0].tolist()
                for r in bparts[0][1]]
        assert [r for r in rows if r] == want
        assert all((np.asarray(r)[-1] >> np.uint32(n % 32)) == 0 for r in bparts[0][1]) if n % 32 else True
        # the compiled host reads the same fileset (--plink-ref-allele a1 for the flipped one) and returns the same rows
        stdout, got = _parse_only_rows(_driver_exe(), [prefix + ".bed"], str(tmp_path / "o"),
                                       extra=["--references", "chr17:41196311:41277499"] + (["--plink-ref-allele", "a1"] if flip else []))
        assert "Matrix size: %d." % n in stdout and got == want
        # the prefix and the .fam name the same fileset; a region that holds nothing gives no rows
        assert np.array_equal(ingest.load_plink(prefix, None, ref_allele="a1" if flip else "a2")[2][0][1], idx)
        empty = ingest.load_plink(prefix + ".fam", ["chr17:1:100"])[2][0]
        assert empty[1].size == 0 and empty[2].tolist() == [0]


def test_plink_sex_and_mitochondrial_chromosome_codes_are_dropped_like_x_y_mt_in_a_vcf(tmp_path):
    """A .bim names X / Y / XY / MT / unplaced as 23 / 24 / 25 / 26 / 0.  The reference's contig rule keeps [a-z]*[0-9]+ only
    (VariantsRDD.scala:103-110), i.e. drops X, Y, MT of a VCF; the same variants written to a PLINK fileset must be dropped
    too, by both hosts, or the two formats of one cohort give different S."""
    from conftest import write_golden_plink
    g = load_golden("pops40")
    offs = g["row_offsets"]
    n = int(g["n_samples"])
    prefix = str(tmp_path / "sex")
    write_golden_plink(g, prefix)
    lines = open(prefix + ".bim").read().splitlines()
    codes = {1: "23", 2: "24", 4: "25", 5: "26", 7: "0", 8: "X", 9: "MT"}
    for k, code in codes.items():
        t = lines[k].split("\t")
        t[0] = code
        lines[k] = "\t".join(t)
    open(prefix + ".bim", "w").write("\n".join(lines) + "\n")
    variants = json.loads(str(g["variants_json"]))
    ids = [str(s) for s in g["callset_ids"]]
    ingest = load_pkg("ingest")
    kind, idx, o = ingest.load_plink(prefix + ".bed", None)[2][0]
    got = [idx[o[k]:o[k + 1]].tolist() for k in range(len(o) - 1)]
    # expected: the carrier rows of the variants that keep an autosomal code, empty rows dropped (VariantsPca.scala:166)
    want = []
    for k, var in enumerate(variants):
        if k in codes:
            continue
        row = sorted(ids.index(c["callSetId"]) for c in var.get("calls", []) if any(a > 0 for a in c["genotype"]))
        if row:
            want.append(row)
    assert got == want and len(got) < len(offs) - 1
    stdout, got_cpp = _parse_only_rows(_driver_exe(), [prefix + ".bed"], str(tmp_path / "o"), extra=["--all-references"])
    assert got_cpp == want and "Matrix size: %d." % n in stdout


def test_plink_reader_refuses_what_it_cannot_read(tmp_path):
    from conftest import write_golden_plink
    ingest = load_pkg("ingest")
    g = load_golden(golden_cases()[0])
    prefix = str(tmp_path / "p")
    write_golden_plink(g, prefix)
    raw = open(prefix + ".bed", "rb").read()
    open(prefix + ".bed", "wb").write(raw[:2] + bytes([0]) + raw[3:])           # sample-major flag
    with pytest.raises(ValueError):
        ingest.load_plink(prefix)
    open(prefix + ".bed", "wb").write(raw[:-1])                                 # truncated
    with pytest.raises(ValueError):
        ingest.load_plink(prefix)
    open(prefix + ".bed", "wb").write(b"BCF" + raw[3:])                         # another format
    with pytest.raises(ValueError):
        ingest.load_plink(prefix)
    import subprocess
    for bad in (raw[:2] + bytes([0]) + raw[3:], raw[:-1], raw + b"\x00", b"BCF" + raw[3:]):   # the compiled host says no too
        open(prefix + ".bed", "wb").write(bad)
        res = subprocess.run([_driver_exe(), "--input-path", prefix + ".bed", "--all-references", "--parse-only"],
                             stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        assert res.returncode != 0
    open(prefix + ".bed", "wb").write(raw)
    res = subprocess.run([_driver_exe(), "--input-path", prefix + ".bed", prefix + ".bed", "--all-references", "--parse-only"],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert res.returncode != 0 and b"needs VCF inputs" in res.stdout
    # the driver front end: one PLINK fileset is a carrier source like a .npz; joins and the AF filter need VCF records
    vp = load_pkg("variants_pca")
    open(prefix + ".bed", "wb").write(raw)
    conf = vp.PcaConf(["--input-path", prefix + ".bed", "--all-references"])
    indexes, names, data = vp.load_dataset(conf)
    assert len(indexes) == int(g["n_samples"]) and data[0][0] == "bed"   # r05: raw rows, decoded on the device

    class Recorder(object):                     # what the front end hands to the engine, without a GPU
        def __init__(self):
            self.bits, self.bed, self.finalized = [], [], False

        def accumulate_bits(self, b):
            self.bits.append(np.array(b))

        def accumulate_plink_bed(self, rows, ref_is_a1=False):
            assert rows.dtype == np.uint8 and rows.flags["C_CONTIGUOUS"]
            self.bed.append((np.array(rows), ref_is_a1))

        def finalize(self):
            self.finalized = True

    driver = vp.VariantsPcaDriver(conf, indexes, names, data)
    rec = Recorder()
    out = vp.calculate_similarity_matrix(driver.getCallsRdd([driver.filterDataset(d) for d in driver.data]),
                                         len(indexes), engine=rec)
    nvar = len(json.loads(str(g["variants_json"])))
    assert out is rec and rec.finalized and not rec.bits and len(rec.bed) == 1 and rec.bed[0][1] is False
    assert np.array_equal(rec.bed[0][0], np.frombuffer(raw[3:], dtype=np.uint8).reshape(nvar, -1))
    # a --references window squeezes the rows outside it out of every block, and blocks without a kept row are skipped
    old_block = vp.PLINK_BLOCK_ROWS
    try:
        vp.PLINK_BLOCK_ROWS = 3
        lo, hi = 41196312 + 7 * 4 - 1, 41196312 + 7 * 11 - 1        # variants 4 .. 10 of write_golden_plink's .bim
        conf2 = vp.PcaConf(["--input-path", prefix + ".bed", "--references", "17:%d:%d" % (lo, hi)])
        i2, n2, d2 = vp.load_dataset(conf2)
        rec2 = Recorder()
        drv2 = vp.VariantsPcaDriver(conf2, i2, n2, d2)
        vp.calculate_similarity_matrix(drv2.getCallsRdd([drv2.filterDataset(d) for d in drv2.data]), len(i2), engine=rec2)
        got = np.concatenate([b for b, _ in rec2.bed])
        assert np.array_equal(got, np.frombuffer(raw[3:], dtype=np.uint8).reshape(nvar, -1)[4:11]) and len(rec2.bed) == 3
    finally:
        vp.PLINK_BLOCK_ROWS = old_block
    # the numpy decode of the same bytes (as_bits) is what the device decode must reproduce: held together on the GPU by
    # tests/test_gpu_multi_engine.py::test_peer_reduction_and_device_side_bed_decode_through_the_c_abi
    with pytest.raises(SystemExit):
        vp.load_dataset(vp.PcaConf(["--input-path", prefix + ".bed", prefix + ".bed", "--all-references"]))


def test_variant_sets_with_the_same_file_stem_keep_distinct_callsets(tmp_path):
    """a/cohort.chr17.vcf + b/cohort.chr17.vcf used to collapse to one set of callset ids in the
    Python host (N = 2 instead of 4, indices out of range).  Both hosts: positional indices, unique ids, same rows."""
    vp = load_pkg("variants_pca")
    os.makedirs(str(tmp_path / "a"))
    os.makedirs(str(tmp_path / "b"))
    body = {"a": ["0|1\t0|0", "1|1\t0|1", "0|0\t0|1"], "b": ["0|0\t1|0", "0|1\t1|1", "1|0\t0|0"]}
    paths = []
    for d in ("a", "b"):
        path = str(tmp_path / d / "cohort.chr17.vcf")
        with open(path, "w") as f:
            f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t%s1\t%s2\n" % (d, d))
            for k, cells in enumerate(body[d]):
                f.write("17\t%d\t.\tA\tG\t.\tPASS\tAF=0.5\tGT\t%s\n" % (41196400 + k, cells))
        paths.append(path)
    conf = vp.PcaConf(["--input-path"] + paths)
    indexes, names, data = vp.load_dataset(conf)
    assert sorted(indexes.values()) == [0, 1, 2, 3] and len(names) == 4
    assert sorted(indexes, key=indexes.get) == ["cohort-0", "cohort-1", "cohort_1-0", "cohort_1-1"]
    assert [names[c] for c in sorted(indexes, key=indexes.get)] == ["a1", "a2", "b1", "b2"]
    drv = vp.VariantsPcaDriver(conf, indexes, names, data)
    rows = sorted(sorted(r) for r in drv.getCallsRdd(data))
    assert rows == [[0, 1, 2, 3], [0, 3], [1, 2]]   # per site: carriers of a ++ carriers of b (index base 2)
    stdout, got = _parse_only_rows(_driver_exe(), paths, str(tmp_path / "o"))
    assert "Matrix size: 4." in stdout
    assert sorted(sorted(r) for r in got) == rows


def test_allele_frequency_filter_and_joins_refuse_carrier_only_inputs(tmp_path):
    """--min-allele-frequency on a .npz / synthetic dataset used to be ignored silently."""
    vp = load_pkg("variants_pca")
    conf = vp.PcaConf(["--synthetic", "50,12,3", "--min-allele-frequency", "0.1"])
    indexes, names, data = vp.load_dataset(conf)
    drv = vp.VariantsPcaDriver(conf, indexes, names, data)
    with pytest.raises(ValueError):
        drv.filterDataset(data[0])
    path = str(tmp_path / "d.npz")
    np.savez(path, callset_ids=np.array(["s-0", "s-1"]), sample_idx=np.array([0, 1], dtype=np.int32),
             row_offsets=np.array([0, 2], dtype=np.int64))
    with pytest.raises(SystemExit):
        vp.load_dataset(vp.PcaConf(["--input-path", path, "--min-allele-frequency", "0.1"]))
    with pytest.raises(SystemExit):
        vp.load_dataset(vp.PcaConf(["--input-path", path, path]))


def test_gzip_path_with_shell_metacharacters_is_just_a_path(tmp_path):
    """VERDICT r01 hygiene: the compiled host used to build `gzip -dc '<path>'` for popen; a quote in the path was a
    shell injection.  gzip is now spawned with the path as one argv entry."""
    import gzip
    odd = str(tmp_path / "it's; touch INJECTED #.vcf.gz")
    with gzip.open(odd, "wt") as f:
        f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tx\ty\n")
        f.write("17\t41196400\t.\tA\tG\t.\tPASS\t.\tGT\t0|1\t1|1\n")
    cwd = os.getcwd()
    os.chdir(str(tmp_path))
    try:
        stdout, got = _parse_only_rows(_driver_exe(), [odd], str(tmp_path / "o"))
    finally:
        os.chdir(cwd)
    assert got == [[0, 1]] and not os.path.exists(str(tmp_path / "INJECTED"))


def test_jni_shim_compiles_against_the_stub_and_covers_every_scala_native(tmp_path):
    """SURVEY 8(f) rank 4 as source (no JDK in the image): jni/pcoa_jni.cpp must compile (here against
    tests/jni_stub/jni.h), export one Java_..._NativePcoa_00024_<name> per @native of scala/.../NativePcoa.scala, and
    forward only to functions include/pcoa.h declares.  The replay program that RUNS the shim is a GPU test."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    scala = open(os.path.join(ROOT, "scala", "com", "google", "cloud", "genomics", "spark", "examples", "NativePcoa.scala")).read()
    natives = set(re.findall(r"@native\s+def\s+(\w+)", scala))
    assert len(natives) >= 14
    obj = str(tmp_path / "pcoa_jni.o")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-c", "-I", os.path.join(ROOT, "tests", "jni_stub"),
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "jni", "pcoa_jni.cpp"), "-o", obj])
    syms = subprocess.check_output(["nm", "--defined-only", obj], universal_newlines=True)
    exported = set(re.findall(r" T Java_com_google_cloud_genomics_spark_examples_NativePcoa_00024_(\w+)", syms))
    assert exported == natives
    undefined = subprocess.check_output(["nm", "--undefined-only", obj], universal_newlines=True)
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "pcoa.h")).read(), flags=re.S)
    declared = set(re.findall(r"\b(pcoa_[a-z0-9_]+)\s*\(", header))
    used = set(re.findall(r" U (pcoa_[a-z0-9_]+)", undefined))
    assert used and used <= declared
    # the Scala host calls only natives that exist
    host = open(os.path.join(ROOT, "scala", "com", "google", "cloud", "genomics", "spark", "examples", "VariantsPcaNative.scala")).read()
    called = set(re.findall(r"NativePcoa\.(\w+)\(", host)) - {"direct", "check"}
    assert called and called <= natives


def test_hot_kernels_do_not_spill_to_scratch():
    """A register spill in a Gram kernel costs an order of magnitude (seen once: 1,632 B/lane of scratch made
    the i8 contraction 45x slower while every parity test stayed green).  hipcc reports it at compile time."""
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "spark-examples_amd", "csrc")
    with tempfile.TemporaryDirectory() as td:
        for src in ("gram_packed.hip", "gram_f32.hip", "eig_lanczos.hip"):
            res = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", os.path.join(ROOT, "include"),
                                  "-I", csrc, "-c", os.path.join(csrc, src), "-o", os.path.join(td, "x.o"),
                                  "-Rpass-analysis=kernel-resource-usage"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                                 universal_newlines=True)
            assert res.returncode == 0, res.stdout[-2000:]
            names = re.findall(r"Function Name: (\S+)", res.stdout)
            scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", res.stdout)]
            assert len(names) == len(scratch) and len(names) >= 2
            vgprs = [int(x) for x in re.findall(r"  VGPRs: (\d+)", res.stdout)]
            assert len(vgprs) == len(names)
            for nm, sc, vg in zip(names, scratch, vgprs):
                # (r05: the large-N mat-vec / row sums spilled while they were being written -- the compiler interleaved
                # four rows -- and the LDS scatter of the carrier lists is new: held to the same rule)
                if any(t in nm for t in ("gram_", "symv_sym_tiles", "rowsums_sym_tiles", "densify_csr_kbits_lds")):
                    assert sc == 0, "%s spills %d bytes/lane" % (nm, sc)
                if "symv_sym_tiles" in nm:
                    assert vg <= 128, "%s: %d VGPRs, four workgroups per CU need <= 128" % (nm, vg)


# ---- the same ingest tests through the AddressSanitizer + UBSan build of the compiled host ------------------
def _sanitizer_exe():
    import subprocess
    hdir = os.path.join(ROOT, "spark-examples_amd", "host")
    if not os.path.exists(os.path.join(ROOT, "spark-examples_amd", "libpcoa_hip.so")):
        pytest.skip("libpcoa_hip.so not built")
    res = subprocess.run(["make", "-s", "-C", hdir, "sanitize"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         universal_newlines=True)
    exe = os.path.join(ROOT, "spark-examples_amd", "variants_pca_driver_san")
    if res.returncode != 0 or not os.path.exists(exe):
        pytest.skip("no sanitizer build on this machine: " + res.stdout[-300:])
    return exe


@pytest.mark.parametrize("which", ["ingest", "goldens_vcf", "goldens_plink", "plink_refusals", "same_stem", "gzip_path", "streamed_join"])
def test_compiled_host_again_under_asan_and_ubsan(which, tmp_path, monkeypatch):
    """Every --parse-only test of this file once more with the sanitizer build in place of the release binary: a heap
    overrun, use-after-free, leak or undefined operation in the VCF / PLINK readers, the joins or the option parsing makes
    the child exit non-zero (abort_on_error / halt_on_error), which those tests assert against."""
    san = _sanitizer_exe()
    me = sys.modules[__name__]
    monkeypatch.setattr(me, "_driver_exe", lambda: san)
    # libpcoa_hip.so pulls in the HIP runtime, whose start-up allocations are not ours to judge: leaks are reported for
    # the host's own frames only through the suppression of everything below the runtime libraries
    supp = tmp_path / "lsan.supp"
    supp.write_text("leak:libamdhip64\nleak:libhsa-runtime64\nleak:librocprofiler\nleak:libamd_comgr\nleak:libdrm\n")
    monkeypatch.setenv("ASAN_OPTIONS", "abort_on_error=1:detect_leaks=1:strict_string_checks=1")
    monkeypatch.setenv("LSAN_OPTIONS", "suppressions=%s:print_suppressions=0" % supp)
    monkeypatch.setenv("UBSAN_OPTIONS", "halt_on_error=1:print_stacktrace=1")
    if which == "ingest":
        test_compiled_host_ingest_matches_python_ingest(tmp_path)
    elif which == "goldens_vcf":
        for name in golden_cases():
            d = tmp_path / ("v_" + name)
            d.mkdir()
            test_vcf_ingest_of_both_hosts_reproduces_the_reference_carrier_rows(name, d)
    elif which == "goldens_plink":
        for name in golden_cases():
            d = tmp_path / ("p_" + name)
            d.mkdir()
            test_plink_fileset_ingest_reproduces_the_reference_carrier_rows(name, d)
    elif which == "plink_refusals":
        test_plink_reader_refuses_what_it_cannot_read(tmp_path)
        d = tmp_path / "sexchrom"
        d.mkdir()
        test_plink_sex_and_mitochondrial_chromosome_codes_are_dropped_like_x_y_mt_in_a_vcf(d)
    elif which == "same_stem":
        test_variant_sets_with_the_same_file_stem_keep_distinct_callsets(tmp_path)
    elif which == "streamed_join":   # r06: spill files, partition-wise join / merge
        for i, (nsets, extra) in enumerate([(2, ["--min-allele-frequency", "0.1"]), (3, ["--join-partitions", "7"])]):
            d = tmp_path / ("j%d" % i)
            d.mkdir()
            test_streamed_join_and_merge_equal_the_in_memory_path(d, nsets, extra)
    else:
        test_gzip_path_with_shell_metacharacters_is_just_a_path(tmp_path)


def test_bench_refuses_counter_traffic_of_another_tree(tmp_path, monkeypatch):
    """roofline.traffic must come from --pmc passes of the tree being benchmarked: profiles/gram_pmc_live.json names its
    tree by source hash (csrc/* + pcoa.h) and bench.pmc_for returns nothing for a file made from other sources."""
    import importlib
    import json
    import shutil
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    lib = importlib.import_module("spark-examples_amd._lib")
    here = lib.source_hash()
    assert len(here) == 16 and here == lib.source_hash()
    (tmp_path / "profiles").mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    rec, src = bench.pmc_for("kbits")
    assert rec == {} and "no profiles/gram_pmc_live.json" in src
    live = tmp_path / "profiles" / "gram_pmc_live.json"
    live.write_text(json.dumps({"source_hash": "0" * 16, "kbits": {"gram_hbm_bytes_per_mvariants": 1.0}}))
    rec, src = bench.pmc_for("kbits")
    assert rec == {} and src.startswith("REFUSED")
    live.write_text(json.dumps({"source_hash": here, "made": "today", "kbits": {"gram_hbm_bytes_per_mvariants": 1.0}}))
    rec, src = bench.pmc_for("kbits")
    assert rec == {"gram_hbm_bytes_per_mvariants": 1.0} and here in src
    # the hash follows the kernel sources: the same files under another root with one byte more give another hash
    pkg_copy = tmp_path / "pkg"
    shutil.copytree(os.path.join(ROOT, "spark-examples_amd", "csrc"), pkg_copy / "p" / "csrc",
                    ignore=shutil.ignore_patterns("*.o", "*.so", "*.hsaco", "*.s"))
    shutil.copytree(os.path.join(ROOT, "include"), pkg_copy / "include")
    monkeypatch.setattr(lib, "_HERE", str(pkg_copy / "p"))
    assert lib.source_hash() == here
    with open(pkg_copy / "p" / "csrc" / "center.hip", "a") as fh:
        fh.write("\n")
    assert lib.source_hash() != here


def test_pmc_live_turns_counter_csvs_into_per_variant_traffic(tmp_path):
    """tools/pmc_live.py: per-process FETCH_SIZE / WRITE_SIZE totals of the two kernel families / the variants the manifest
    names, reads doubled (gfx950 tallies wide reads at half), KiB -> bytes; the file carries the manifest's source hash."""
    import csv
    import json
    import subprocess
    out = tmp_path / "tag"
    rows = {"FETCH_SIZE": [("void pcoa::(anonymous namespace)::pack_kbits_ring_kernel<8, 0, 0>(float const*)", 1, 1000.0),
                           ("void pcoa::(anonymous namespace)::pack_kbits_ring_kernel<8, 0, 0>(float const*)", 2, 3000.0),
                           ("void pcoa::(anonymous namespace)::gram_kbits_kernel<3, 2, 2>(signed char const*)", 3, 500.0),
                           ("void at::native::some_other_kernel()", 4, 1e9)],
            "WRITE_SIZE": [("void pcoa::(anonymous namespace)::pack_kbits_kernel<float, 4, true, false>(float const*)", 1, 100.0),
                           ("void pcoa::(anonymous namespace)::gram_kbits_w4_kernel<4, 0, 69, 0>(signed char const*)", 2, 50.0)]}
    for tag in ("pipe", "serial"):
        for c, rs in rows.items():
            sub = out / ("pmclive_%s_%s" % (tag, c)) / "host"
            sub.mkdir(parents=True)
            with open(sub / "pmc_counter_collection.csv", "w", newline="") as fh:
                w = csv.writer(fh)
                w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
                for name, disp, val in rs:
                    w.writerow([disp, name, c, val])
            (out / ("pmclive_%s_%s.json" % (tag, c))).write_text(json.dumps(
                {"pmc_manifest": {"source_hash": "abcd" * 4, "fp32_variants_through_the_engine": 2000000}}))
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_live.py"), str(out)], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, universal_newlines=True)
    assert res.returncode == 0, res.stdout
    rec = json.load(open(out / "gram_pmc_live.json"))
    assert rec["source_hash"] == "abcd" * 4
    for key in ("kbits", "kbits_standalone"):
        k = rec[key]
        assert k["pack_read_bytes_per_mvariants"] == 2 * 4000.0 * 1024 / 2.0      # 2 x KiB x 1024 / 2e6 variants x 1e6
        assert k["gram_read_bytes_per_mvariants"] == 2 * 500.0 * 1024 / 2.0
        assert k["pack_write_bytes_per_mvariants"] == 100.0 * 1024 / 2.0 and k["gram_write_bytes_per_mvariants"] == 50.0 * 1024 / 2.0
        assert k["pack_hbm_bytes_per_mvariants"] == k["pack_read_bytes_per_mvariants"] + k["pack_write_bytes_per_mvariants"]
        assert k["pack_dispatches"] == 1 and k["gram_dispatches"] == 1   # (the last pass read: WRITE_SIZE)


@pytest.mark.parametrize("nsets,extra", [(2, []), (3, []), (2, ["--min-allele-frequency", "0.1"]), (3, ["--join-partitions", "1"]),
                                         (2, ["--join-partitions", "7"])])
def test_streamed_join_and_merge_equal_the_in_memory_path(tmp_path, nsets, extra):
    """r06 (VERDICT r05 Missing 6): joins / merges of the compiled host are streamed -- every variant set once through the parser
    threads into hash-partitioned spill files (getVariantKey), then one key partition at a time is joined / merged (the reference's
    join is a shuffle, VariantsPca.scala:115-148).  --parse-only runs exactly that without a GPU: the RDD[Seq[Int]] rows must be
    the in-memory path's (--no-stream), as a multiset -- the order of the joined rows is the partitions', and S does not care."""
    rng = np.random.default_rng(17 + nsets)
    bases = "ACGT"
    paths = []
    for d in range(nsets):
        nsamp = (6, 5, 4)[d]
        path = str(tmp_path / ("set%d.vcf" % d))
        with open(path, "w") as f:
            f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" +
                    "\t".join("D%dS%d" % (d, i) for i in range(nsamp)) + "\n")
            for k in range(400):
                if d == 1 and k % 5 == 0:
                    continue                                        # site missing from set 1: not joined
                ref = bases[k % 4]
                alt = bases[(k + 1 + (d == 2 and k % 7 == 0)) % 4]   # a different ALT in set 2 now and then: another key
                gts = "\t".join("%d|%d" % (rng.random() < 0.3, rng.random() < 0.3) for _ in range(nsamp))
                f.write("17\t%d\t.\t%s\t%s\t.\tPASS\tAF=%.3f\tGT\t%s\n" % (41196400 + 3 * k, ref, alt, rng.uniform(0, 0.3), gts))
                if k % 97 == 0:                                      # the same key twice inside one file
                    f.write("17\t%d\t.\t%s\t%s\t.\tPASS\tAF=0.2\tGT\t%s\n" % (41196400 + 3 * k, ref, alt, gts))
        paths.append(path)
    drv = _driver_exe()
    out_s, rows_s = _parse_only_rows(drv, paths, str(tmp_path / "s"), extra=["--all-references"] + extra)
    out_m, rows_m = _parse_only_rows(drv, paths, str(tmp_path / "m"), extra=["--all-references", "--no-stream"] +
                                     [e for e in extra if not e.startswith("--join") and not e.isdigit()])
    assert "Streamed join" in out_s and "Streamed join" not in out_m
    assert len(rows_s) > 50 and sorted(sorted(r) for r in rows_s) == sorted(sorted(r) for r in rows_m)
    assert not [f for f in os.listdir(os.environ.get("TMPDIR", "/tmp")) if f.startswith("pcoa_join_")]   # spill files removed
