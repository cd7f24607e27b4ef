"""End-to-end check of the PLINK input path on a GPU: one golden fixture as VCF and as .bed/.bim/.fam through the Python
driver; S must equal the reference's similarity matrix and the two outputs each other (profiles/r02ac_plink_e2e.txt)."""
import importlib, io, os, sys, contextlib, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden, write_golden_vcf, write_golden_plink
vp = importlib.import_module("spark-examples_amd.variants_pca")
g = load_golden("tile260")
d = tempfile.mkdtemp()
write_golden_vcf(g, os.path.join(d, "cohort.vcf")); write_golden_plink(g, os.path.join(d, "cohort"))
outs = []
for path in ("cohort.vcf", "cohort.bed"):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        vp.main(["--input-path", os.path.join(d, path), "--all-references", "--dump-similarity", os.path.join(d, path + ".s")])
    outs.append(buf.getvalue())
import numpy as np
s1 = np.fromfile(os.path.join(d, "cohort.vcf.s"), dtype="<i8"); s2 = np.fromfile(os.path.join(d, "cohort.bed.s"), dtype="<i8")
print("S equal:", bool(np.array_equal(s1, s2)), "== golden:", bool(np.array_equal(s1.reshape(260, 260), g["similarity"])))
print("stdout equal:", outs[0] == outs[1], len(outs[0].splitlines()), "lines")
