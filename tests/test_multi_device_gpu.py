"""Several GPUs behind one handle of the C ABI, one process (csrc/multi_device.cpp, include/mi355lz4.h): the batch is cut into
one contiguous block range per device.  Rehearsed with 1, 2 and 3 engines on the ONE GPU of the test box (the device list
names device 0 again and again): bit-exact against the single-engine call, the per-block results in the caller's order,
ragged and empty blocks, a failing block's code in its place, the block-capacity packing across ranges.
(No multi-GPU node was available: the path is unmeasured on hardware with more than one device.)"""
import random

import pytest

pytestmark = pytest.mark.gpu


def _blocks(oracle, rng, n):
    out = []
    for i in range(n):
        kind = rng.choice(["lzsynth", "text", "random"])
        ln = rng.choice([0, 1, 13, 100, 4096, 65536, 65536, 65536, 70001, 200000])
        out.append(oracle.gen(kind, 1, max(ln, 1), first_block=500 + i)[:ln].tobytes())
    return out


@pytest.mark.parametrize("ndev", [1, 2, 3])
def test_multi_handle_matches_single_engine(slz4, engine, oracle, ndev):
    rng = random.Random(40 + ndev)
    m = slz4.MultiEngine([0] * ndev)
    try:
        for blocks in (_blocks(oracle, rng, 37), [oracle.gen("text", 1, 65536, first_block=k).tobytes() for k in range(300)],
                       [b"x"], [b""], _blocks(oracle, rng, 2)):
            raw = b"".join(blocks)
            fr1, flen1 = engine.compress_batch(blocks)
            frm, flenm = m.compress_batch(blocks)
            # every range is an ordinary call of its engine: the same blocks in, the same framed bytes out (small batches are cut
            # into segments by several waves and their bytes depend on the batch: compare sizes there, bytes through the decoders)
            assert len(flenm) == len(blocks)
            out, blen = m.decompress_batch(frm)
            assert out == raw and blen == [len(b) for b in blocks]
            out, blen = engine.decompress_batch(frm)
            assert out == raw
            out, blen = m.decompress_batch(fr1)
            assert out == raw and blen == [len(b) for b in blocks]
            # ... and the oracle decodes every block of the multi handle's stream
            pos = 0
            for b, f in zip(blocks, flenm):
                assert oracle.decompress_block(frm[pos + 8:pos + f], len(b)) == (len(b), b)
                pos += f
            assert pos == len(frm)
    finally:
        m.close()


def test_multi_handle_reports_a_failing_block_in_its_place(slz4, engine, oracle):
    blocks = [oracle.gen("text", 1, 65536, first_block=k).tobytes() for k in range(12)]
    fr, flen = engine.compress_batch(blocks)
    pos = sum(flen[:7])                                                        # block 7 (in the second of two ranges / third of three)
    for start in range(300, 6000, 97):
        bad = bytearray(fr)
        bad[pos + 8 + start:pos + 8 + start + 64] = b"\xff" * 64
        want = engine.decompress_batch(bytes(bad), raise_on_block_error=False)[1]
        if want[7] < 0:
            break
    assert want[7] < 0 and all(w == 65536 for i, w in enumerate(want) if i != 7)
    for ndev in (2, 3):
        m = slz4.MultiEngine([0] * ndev)
        try:
            with pytest.raises(slz4.LZ4Error):
                m.decompress_batch(bytes(bad))
            out, blen = m.decompress_batch(bytes(bad), raise_on_block_error=False)
            assert blen == want
        finally:
            m.close()


def test_multi_handle_packs_blocks_that_decode_short(slz4, engine, oracle):
    """BlockMax64KB framing (4-byte headers): every block asks for 64 KiB of room and decodes to less -- the ranges' outputs
    are packed back to back like the single engine's."""
    blocks = [oracle.gen("text", 1, 1000 + 37 * k, first_block=k).tobytes() for k in range(50)]
    fr, _ = engine.compress_batch(blocks, header_kind=4)
    want = engine.decompress_batch(fr, header_kind=4, fixed_uncomp=65536)
    m = slz4.MultiEngine([0, 0, 0])
    try:
        assert m.decompress_batch(fr, header_kind=4, fixed_uncomp=65536) == want
        assert want[0] == b"".join(blocks)
    finally:
        m.close()
