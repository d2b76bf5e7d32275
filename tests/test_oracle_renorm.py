"""LZ4_renormDictT (reference cbits/lz4.c:1545-1562, restated in oracle/lz4_oracle.c `renorm`): once more than 2 GiB
have gone through ONE compression context, the table's offsets are rebased.  The GPU path has no running offset
(positions are block-relative), so the restatement is exercised where it lives: the oracle's stream compressor and the
compiled reference are fed the same 2 GiB + 64 MiB -- zeros, with text around the point where the rebase happens, so
that the table is full of live entries when it does -- and must write the same bytes; the oracle's decoder must give
the input back."""
import numpy as np
import pytest

from oracle.oracle import Oracle, Reference, have_reference

BL = 65536


@pytest.mark.skipif(not have_reference(), reason="oracle/_ref (the compiled reference) is not built here")
def test_renorm_after_2gib_matches_reference():
    orc, ref = Oracle(), Reference()
    n_blocks = (2 << 30) // BL + 1024                     # 2 GiB + 64 MiB through one context
    data = np.zeros(n_blocks * BL, dtype=np.uint8)
    # live text from 32 MiB in front of the 2 GiB mark to the end
    t0 = (2 << 30) // BL - 512
    text = orc.gen("text", 256, BL, first_block=11)
    for b in range(t0, n_blocks):
        k = (b - t0) % 256
        data[b * BL:(b + 1) * BL] = text[k * BL:(k + 1) * BL]
    orc.lib.orc_debug_renorms.restype = __import__("ctypes").c_long
    before = orc.lib.orc_debug_renorms()
    a = orc.frame_compress(data, BL, 1, 8, True)
    assert orc.lib.orc_debug_renorms() == before + 1          # the rebase has really run, once
    b = ref.frame_compress(data, BL, 1, 8, True)
    assert len(a) == len(b) and a == b
    # and the tail decodes back (the blocks behind the rebase reference the blocks in front of it)
    back = orc.frame_decompress(a, n_blocks * BL, 8, BL, True)
    assert len(back) == n_blocks * BL
    tail = np.frombuffer(back, dtype=np.uint8)[t0 * BL:]
    assert np.array_equal(tail, data[t0 * BL:])
