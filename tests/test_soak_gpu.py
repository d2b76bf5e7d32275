"""A short run of scripts/soak_decode.py: randomized data shapes (sequence sizes around the 6 / 8 nodes-per-lane switch, deep
dependency chains, self-overlapping matches, long literal runs, ragged blocks) through decoder variants 1, 2 and 3, on
reference-written and engine-written independent blocks and on the reference's linked stream under the run walker and the
pointer pass; every output must equal the input (the reference's own tests are such round-trip properties, test/Main.hs:57-306)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_decoder_soak(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "soak_decode.py"), "12", str(seed)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "soak ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
