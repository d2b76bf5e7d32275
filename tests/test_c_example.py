"""examples/roundtrip.c: the batched C ABI used from plain C.  Without a GPU: the public headers compile as C99
(-pedantic) and as C++17, the example links against the library and fails loudly (no CPU codec).  With one: it
round-trips independent and linked blocks."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "streamly-lz4_amd", "lib")


def _build(tmp_path):
    exe = str(tmp_path / "roundtrip")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "roundtrip.c"), "-L", LIBDIR, "-lmi355lz4",
                           "-Wl,-rpath," + LIBDIR, "-o", exe])
    return exe


def test_headers_are_c_and_cxx(tmp_path):
    for std, comp, name in (("-std=c99", "gcc", "t.c"), ("-std=c11", "gcc", "t.c"), ("-std=c++17", "g++", "t.cpp")):
        srcs = '#include "mi355lz4.h"\n#include "lz4.h"\nint main(void) { return MI355LZ4_VERSION > 0 ? 0 : 1; }\n'
        if comp == "g++":
            srcs = '#include "streamly_lz4.hpp"\n' + srcs
        p = tmp_path / name
        p.write_text(srcs)
        subprocess.check_call([comp, std, "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-I",
                               os.path.join(ROOT, "include"), str(p)])


def test_example_fails_loudly_without_a_device(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    exe = _build(tmp_path)
    r = subprocess.run([exe, "4"], capture_output=True, text=True)
    assert r.returncode == 2 and "no HIP device" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("linked", [0, 1])
def test_example_round_trips(tmp_path, linked):
    exe = _build(tmp_path)
    r = subprocess.run([exe, "300", str(linked)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "round trip ok" in r.stdout, (r.stdout, r.stderr)
    assert ("linked" if linked else "independent") in r.stdout
