"""The Haskell shim (haskell-shim/Streamly/Internal/LZ4/GPU.hs) cannot be compiled here (no GHC): its hand-written
`foreign import` lines are checked mechanically against include/mi355lz4.h instead (scripts/check_haskell_ffi.py).
Reference imports this shim replaces: src/Streamly/Internal/LZ4.hs:105-143."""
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import check_haskell_ffi as ffi  # noqa: E402

SHIM = os.path.join(ROOT, "haskell-shim", "Streamly", "Internal", "LZ4", "GPU.hs")
INC = os.path.join(ROOT, "include")


def test_shim_matches_header():
    rc, msg = ffi.check(SHIM, INC)
    assert rc == 0, msg
    names = [c for _h, c, *_ in ffi.parse_imports(open(SHIM).read())]
    # the two calls that replace the reference's per-block primitives must be among the checked ones
    assert "mi355lz4_compress_batch" in names and "mi355lz4_decompress_batch" in names


def _mutated_header(tmp_path, pattern, repl):
    inc = tmp_path / "include"
    shutil.copytree(INC, inc)
    h = inc / "mi355lz4.h"
    text = h.read_text()
    new, n = re.subn(pattern, repl, text, count=1, flags=re.S)
    assert n == 1
    h.write_text(new)
    return str(inc)


def test_added_argument_in_header_is_caught(tmp_path):
    """An argument added to mi355lz4_decompress_batch without touching the shim must fail the check."""
    inc = _mutated_header(tmp_path, r"(int mi355lz4_decompress_batch\(mi355lz4_ctx \*ctx,)", r"\1 int newFlag,")
    rc, msg = ffi.check(SHIM, inc)
    assert rc == 1 and "mi355lz4_decompress_batch" in msg


def test_width_change_in_header_is_caught(tmp_path):
    """size_t inLen -> int inLen: same arity, different width."""
    inc = _mutated_header(tmp_path, r"(int mi355lz4_decompress_batch\(mi355lz4_ctx \*ctx, const uint8_t \*framedIn,) size_t inLen",
                          r"\1 int inLen")
    rc, msg = ffi.check(SHIM, inc)
    assert rc == 1


def test_wrong_type_in_shim_is_caught(tmp_path):
    """Ptr Int32 -> Ptr CSize on the Haskell side (a 4-byte array read as 8-byte elements)."""
    text = open(SHIM).read()
    m = re.search(r"c_compressBatch\s*::[^\n]*\n[^\n]*Ptr Int32", text)
    assert m
    bad = text[: m.end() - len("Ptr Int32")] + "Ptr CSize" + text[m.end():]
    p = tmp_path / "GPU.hs"
    p.write_text(bad)
    rc, _ = ffi.check(str(p), INC)
    assert rc == 1


def test_type_mapping():
    assert ffi.hs_type_to_c("Ptr (Ptr Word8)") == "uint8_t * *"
    assert ffi.hs_type_to_c("Ptr C_Engine") == "mi355lz4_ctx *"
    assert ffi.split_arrows("Ptr (Ptr Word8) -> CInt -> IO CInt") == ["Ptr (Ptr Word8)", "CInt", "IO CInt"]
