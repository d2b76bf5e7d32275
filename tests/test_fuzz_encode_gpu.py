"""Encoder fuzz on structured inputs: runs, short periods, splices of text / random / repeats, every
length class, several accelerations -- one ragged batch per acceleration through the host API.  Each
block must decode through the ORACLE (the reference's decoder) to exactly its input, stay within
LZ4_compressBound, and survive the GPU decoders too.  ENC_FUZZ_CASES raises the count."""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _make(rng, oracle, t):
    n = rng.choice([0, 1, 5, 12, 13, 14, 40, 63, 64, 65, 127, 128, 129, 500, 1000, 4095, 4096, 4097, 20000, 65535, 65536,
                    65537, 70000, 131073]) if t % 3 == 0 else rng.randrange(0, 6000)
    mode = rng.randrange(8)
    if n == 0:
        return b""
    if mode == 0:                                                   # one run
        return bytes([rng.randrange(256)]) * n
    if mode == 1:                                                   # short period
        k = rng.randrange(1, 70)
        pat = bytes(rng.randrange(256) for _ in range(k))
        return (pat * (n // k + 1))[:n]
    if mode == 2:                                                   # runs of random lengths
        out = bytearray()
        while len(out) < n:
            out += bytes([rng.randrange(4)]) * rng.randrange(1, 300)
        return bytes(out[:n])
    if mode == 3:                                                   # 0/1 bytes (the reference tests' generator)
        return bytes(rng.getrandbits(1) for _ in range(n))
    if mode == 4:                                                   # splice of generators
        parts = bytearray()
        while len(parts) < n:
            kind = rng.choice(["lzsynth", "text", "random"])
            parts += oracle.gen(kind, 1, rng.randrange(1, 3000), first_block=rng.randrange(1 << 20)).tobytes()
        return bytes(parts[:n])
    if mode == 5:                                                   # a block repeated with single-byte edits
        base = bytearray(oracle.gen("text", 1, max(1, n // 4), first_block=t).tobytes())
        out = bytearray()
        while len(out) < n:
            b = bytearray(base)
            b[rng.randrange(len(b))] ^= 0x20
            out += b
        return bytes(out[:n])
    if mode == 6:                                                   # long match right at the end / tiny tail
        half = oracle.gen("random", 1, max(1, n // 2), first_block=t).tobytes()
        return (half + half)[:n]
    return oracle.gen("lzsynth", 1, n, first_block=t, lit_max=rng.choice([1, 4, 16, 64]),
                      off_max=rng.choice([1, 8, 300, 65535])).tobytes()


@pytest.mark.parametrize("accel", [1, 2, 9, 400])
def test_encode_fuzz_structured(engine, oracle, accel):
    n_cases = int(os.environ.get("ENC_FUZZ_CASES", "400"))
    rng = random.Random(1000 + accel)
    blocks = [_make(rng, oracle, t) for t in range(n_cases)]
    fr, flen = engine.compress_batch(blocks, accel=accel)
    assert len(fr) == sum(flen)
    pos = 0
    for i, (b, f) in enumerate(zip(blocks, flen)):
        c = int.from_bytes(fr[pos:pos + 4], "little")
        assert c == f - 8 and int.from_bytes(fr[pos + 4:pos + 8], "little") == len(b)
        assert 0 < c <= oracle.compress_bound(len(b)), (i, len(b), c)
        code, out = oracle.decompress_block(fr[pos + 8:pos + f], len(b))
        assert code == len(b) and out == b, (i, len(b), code)
        pos += f
    for dec in (2, 1):
        engine.set_decoder(dec)
        try:
            out, blen = engine.decompress_batch(fr)
        finally:
            engine.set_decoder(0)
        assert blen == [len(b) for b in blocks] and out == b"".join(blocks)


@pytest.mark.parametrize("accel", [1, 3, 400])
def test_encode_fuzz_linked(engine, oracle, accel):
    """The same ragged batch compressed as ONE linked stream (previous block = dictionary whenever it is non-empty and
    lies directly in front): every block decodes through the oracle's linked decoder, block by block with the output
    of the last non-empty block as dictionary, to exactly its input; the GPU's linked decode agrees; and blocks that
    repeat their predecessor's content do reach into it."""
    n_cases = int(os.environ.get("ENC_FUZZ_CASES", "400"))
    rng = random.Random(7000 + accel)
    blocks = [_make(rng, oracle, t) for t in range(n_cases)]
    for t in range(5, n_cases, 9):                                  # neighbours that share content
        blocks[t] = (blocks[t - 1][-3000:] + blocks[t])[: max(len(blocks[t]), 64)]
    engine.set_linked_compress(True)
    try:
        fr, flen = engine.compress_batch(blocks, accel=accel)
    finally:
        engine.set_linked_compress(False)
    assert len(fr) == sum(flen)
    pos, dict_bytes, reached_back = 0, None, 0
    for i, (b, f) in enumerate(zip(blocks, flen)):
        c = int.from_bytes(fr[pos:pos + 4], "little")
        assert c == f - 8 and int.from_bytes(fr[pos + 4:pos + 8], "little") == len(b)
        assert 0 < c <= oracle.compress_bound(len(b)), (i, len(b), c)
        code, out = oracle.decompress_block(fr[pos + 8:pos + f], len(b), dict_bytes)
        assert code == len(b) and out == b, (i, len(b), code)
        if dict_bytes and oracle.decompress_block(fr[pos + 8:pos + f], len(b))[0] != len(b):
            reached_back += 1
        if len(b) > 0:
            dict_bytes = b
        pos += f
    assert reached_back >= n_cases // 20
    out, blen = engine.decompress_batch(fr, linked=True)
    assert blen == [len(b) for b in blocks] and out == b"".join(blocks)
