"""The shelved experiments (decoder variant 3: token lists by a pass of their own) are not in the shipped
library (`make lib`); `make lib-exp` compiles them into streamly-lz4_amd/lib/libmi355lz4_exp.so.  This test runs the
decoder-parametrized parity tests -- oracle streams, malformed blocks with the reference's exact codes, fuzz, huge
length fields -- on that build with variant 3 only, in ONE child process (the library is chosen at import time)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXP = os.path.join(ROOT, "streamly-lz4_amd", "lib", "libmi355lz4_exp.so")


def test_shipped_library_has_no_experiments(slz4, engine):
    if os.environ.get("MI355LZ4_LIB"):
        pytest.skip("another library was chosen by MI355LZ4_LIB")
    assert not slz4.Engine.has_experiments()
    with pytest.raises(Exception):
        engine.set_decoder(3)
    engine.set_decoder(0)


def test_decoder3_parity_on_experiment_build():
    if os.environ.get("MI355LZ4_TEST_ONLY_DECODER"):
        pytest.skip("this IS the child run")
    assert os.path.exists(EXP), "make lib-exp (or __graft_entry__.build()) builds %s" % EXP
    env = dict(os.environ, MI355LZ4_LIB=EXP, MI355LZ4_TEST_ONLY_DECODER="3")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_parity_gpu.py"), "-x", "-q", "-m", "gpu",
                        "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    tail = r.stdout[-1500:]
    assert r.returncode == 0 and " passed" in tail, (tail, r.stderr[-800:])
