"""Soak test of the linked-stream decode (run-in decode with tiny pieces and run-ins, pointer pass, their fallbacks) against the oracle:
many random streams with random corruptions, ragged block sizes, random segment and piece sizes; every block's result
(size or the reference's negative code) and every decoded byte must be the oracle's linked decode's.  200 trials by
default (a second); LINKED_SOAK=<trials> LINKED_SOAK_SEED=<seed> for a real soak after a change to the linked paths
(round 5: 8 seeds x 3000 trials; it found a piece trusting the stand-in's result where no dictionary was in force, and
pieces finishing behind a broken block)."""
import os
import random

import pytest

from test_parity_gpu import _decode_streams, split_blocks

pytestmark = pytest.mark.gpu
TRIALS = int(os.environ.get("LINKED_SOAK", "200"))


@pytest.mark.skipif(TRIALS <= 0, reason="soak test: set LINKED_SOAK=<trials>")
def test_linked_soak(engine, oracle, monkeypatch):
    rng = random.Random(int(os.environ.get("LINKED_SOAK_SEED", "1")))
    for trial in range(TRIALS):
        kind = rng.choice(["text", "lzsynth", "text"])
        bl = rng.choice([65536, 65536, 32768, 16384, 4096, 1000])
        nblk = rng.randint(2, 24)
        d = oracle.gen(kind, nblk, 65536, first_block=rng.randrange(1 << 20)).tobytes()[: nblk * bl]
        shape = rng.random()
        if shape < 0.15:
            pat = d[: rng.randint(1, 5000)]
            d = (pat * (len(d) // len(pat) + 1))[: len(d)]
        elif shape < 0.25:
            d = d[:bl] * nblk
        elif shape < 0.3:
            d = bytes(len(d))
        elif shape < 0.36:
            # noise with a period just below the reach of an offset: every block is made of the block before it, a missing
            # dictionary is never forgotten (the run-in decode's pieces are all to be redone, in a chain)
            pat = random.Random(rng.getrandbits(32)).randbytes(rng.randint(40000, 65535))
            d = (pat * (len(d) // len(pat) + 1))[: len(d)]
        fr = bytearray(oracle.frame_compress(d, bl, rng.choice([1, 1, 3, 50]), 8, True))
        blocks = split_blocks(bytes(fr))
        for _ in range(rng.choice([0, 0, 1, 1, 2, 5])):
            bi = rng.randrange(0, len(blocks))
            start = sum(len(b) for b in blocks[:bi]) + 8
            pos = start + rng.randrange(max(1, len(blocks[bi]) - 8))
            fr[pos] = rng.randrange(256) if rng.random() < 0.5 else fr[pos] ^ (1 << rng.randrange(8))
        monkeypatch.setenv("MI355LZ4_LINKED_PTR_BLOCKS", str(rng.choice([1, 2, 3, 7, 4096])))
        monkeypatch.setenv("MI355LZ4_LINKED_POOL_BLOCKS", str(rng.choice([2, 5, 16384, 16384])))
        if rng.random() < 0.4:
            # the run-in decode in front of all that (pieces that are redone, chain and run out of rounds; corrupted blocks
            # send the span to the passes above): the reference's codes and bytes all the same
            monkeypatch.setenv("MI355LZ4_LINKED_RUNIN", "1")
            monkeypatch.setenv("MI355LZ4_LINKED_RUNIN_PIECE", str(rng.choice([1, 2, 3, 5])))
            monkeypatch.setenv("MI355LZ4_LINKED_RUNIN_BLOCKS", str(rng.choice([1, 1, 2, 3, 11])))
            monkeypatch.setenv("MI355LZ4_LINKED_RUNIN_SPIN", str(rng.choice([0, 10000, 10000])))
            monkeypatch.setenv("MI355LZ4_LINKED_RUNS", "0")
        else:
            monkeypatch.delenv("MI355LZ4_LINKED_RUNIN", raising=False)
            monkeypatch.delenv("MI355LZ4_LINKED_RUNIN_PIECE", raising=False)
            monkeypatch.delenv("MI355LZ4_LINKED_RUNIN_BLOCKS", raising=False)
            monkeypatch.delenv("MI355LZ4_LINKED_RUNIN_SPIN", raising=False)
            monkeypatch.delenv("MI355LZ4_LINKED_RUNS", raising=False)
        dict_bytes, eres, eouts = None, [], []
        for b in split_blocks(bytes(fr)):
            cap = int.from_bytes(b[4:8], "little")
            if cap < 0 or cap > 1 << 20:
                cap = 0
            code, dec = oracle.decompress_block(b[8:], cap, dict_bytes)
            eres.append(code)
            eouts.append(dec if code >= 0 else None)
            if code > 0:
                dict_bytes = dec
        mode = rng.choice(["one", "one", "streams"])
        frs = [bytes(fr)] if mode == "one" else [bytes(fr), bytes(fr)]
        out, res, ulen, first = _decode_streams(engine, frs, mode)
        reps = len(frs)
        assert res == eres * reps, (trial, [(i, a, b) for i, (a, b) in enumerate(zip(res, eres * reps)) if a != b][:6], {k: os.environ.get(k) for k in os.environ if k.startswith("MI355LZ4_LINKED")}, mode, bl, nblk)
        o = 0
        for r in range(reps):
            for j, e in enumerate(eouts):
                if e is not None:
                    assert out[o:o + len(e)] == e, (trial, r, j)
                o += ulen[r * len(eouts) + j]
