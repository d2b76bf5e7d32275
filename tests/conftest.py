import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "streamly-lz4_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _decoders():
    """Decoder variants the loaded library has: 1 (sequence at a time), 2 (lane-parallel, one wavefront per block), 4 (one
    workgroup per block, also as a linked call's first pass) and, in the experiment build only (make lib-exp, MI355LZ4_LIB
    pointing at it), 3 (token lists).  MI355LZ4_TEST_ONLY_DECODER narrows the list."""
    try:
        import streamly_lz4_amd as S
        ds = [1, 2, 3, 4] if S.Engine.has_experiments() else [1, 2, 4]
    except Exception:
        ds = [1, 2, 4]
    only = os.environ.get("MI355LZ4_TEST_ONLY_DECODER")
    if only:
        ds = [d for d in ds if d == int(only)]
    return ds


DECODERS = _decoders()


def pytest_collection_modifyitems(config, items):
    # tests/test_experiment_build_gpu.py re-runs the decoder-parametrized tests on the experiment build, variant 3 only
    if os.environ.get("MI355LZ4_TEST_ONLY_DECODER"):
        keep = [it for it in items if "decoder" in getattr(it, "fixturenames", ()) and hasattr(it, "callspec") and "decoder" in it.callspec.params]
        drop = [it for it in items if it not in keep]
        if drop:
            config.hook.pytest_deselected(items=drop)
            items[:] = keep


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def reference():
    """The real reference codec; only present where oracle/_ref was built."""
    from oracle.oracle import Reference, have_reference
    if not have_reference():
        pytest.skip("oracle/_ref not built (reference sources absent)")
    return Reference()


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def linked_golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "linked_stream.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def slz4():
    """The product binding.  Import fails loudly when libmi355lz4.so is missing."""
    import streamly_lz4_amd
    return streamly_lz4_amd


@pytest.fixture(scope="session")
def engine(slz4):
    """GPU engine: only requested by gpu-marked tests.  No fallback: creation must succeed."""
    eng = slz4.Engine(0)
    yield eng
    eng.close()
