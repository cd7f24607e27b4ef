"""A short soak of the co-resident pipeline (GPU only): tools/soak_pipeline.py -- steps of 10^6 resident variants as fp32, uint8,
bitsets and mixed, S bit for bit against a multiple of a NO_PIPELINE engine's result.  The round's long run (4 x 20,000 steps =
8e10 variants, entries past 2^31) is profiles/r03zp_soak_pipeline.txt."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_pipeline_is_exact_over_hundreds_of_steps_in_every_input_format():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak_pipeline.py"), os.environ.get("PCOA_SOAK_STEPS", "300")],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=900)
    assert res.returncode == 0 and "SOAK ok" in res.stdout, res.stdout[-3000:]
    # and the pipeline was what ran (bitset tiles run their transpose and their contraction in series since r05)
    for line in res.stdout.splitlines():
        if "pipelined launches" in line:
            piped = int(line.split("pipelined launches")[1].split(",")[0])
            assert (piped == 0) if line.startswith("bitsets") else (piped > 100), line
