"""Development aid: randomized differential soak of the decoders (variants 1, 2, 4 and, in the experiment build, 3) against the input, over data shapes that
move the lane-parallel kernel between its forms: sequence sizes around the 6 / 8 nodes-per-lane switch, deep dependency
chains (short offsets), self-overlapping matches, long literal runs, tiny and huge blocks, ragged last blocks.
    python scripts/soak_decode.py [seconds] [seed]"""
import os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import numpy as np
import streamly_lz4_amd as S
from oracle.oracle import Oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = random.Random(seed)
o = Oracle()
eng = S.Engine(0)


def synth(n, lit_max, off_max, mlen_max, alphabet):
    """literal runs of 1..lit_max symbols from `alphabet`, matches of 4..mlen_max at offsets 1..off_max"""
    out = bytearray()
    r = random.Random(rng.getrandbits(32))
    while len(out) < n:
        L = r.randint(1, lit_max)
        out += bytes(r.choice(alphabet) for _ in range(L))
        if len(out) >= n:
            break
        M = r.randint(4, mlen_max)
        lim = min(len(out), off_max)
        off = r.randint(1, lim)
        for _ in range(M):
            out.append(out[-off])
    return bytes(out[:n])


t0 = time.time()
cases = 0
while time.time() - t0 < budget:
    bl = rng.choice([700, 4096, 20000, 65536, 65536, 65536, 100001, 262144])
    nb = rng.randint(1, 6)
    lit_max = rng.choice([1, 2, 3, 8, 16, 40, 300])
    off_max = rng.choice([1, 3, 7, 40, 300, 2048, 20000, 65535])
    mlen_max = rng.choice([4, 6, 12, 30, 64, 300])
    alphabet = bytes(range(32, 32 + rng.choice([2, 16, 64])))
    total = bl * nb - rng.choice([0, 0, 1, 17, bl // 2])
    data = synth(total, lit_max, off_max, mlen_max, alphabet)
    accel = rng.choice([1, 1, 1, 4, 64])
    fr_ref = o.frame_compress(data, bl, accel, 8, False)                 # reference-written independent blocks
    comp_gpu = eng.compress_batch([data[i:i + bl] for i in range(0, len(data), bl)], accel=accel)[0]
    for dv in ((1, 2, 3, 4) if S.Engine.has_experiments() else (1, 2, 4)):
        eng.set_decoder(dv)
        out, blen = eng.decompress_batch(fr_ref)
        assert out == data, ("ref stream", dv, bl, nb, lit_max, off_max, mlen_max, accel, cases)
        if comp_gpu is not None:
            out, blen = eng.decompress_batch(comp_gpu)
            assert out == data, ("gpu stream", dv, bl, nb, lit_max, off_max, mlen_max, accel, cases)
    eng.set_decoder(0)
    # the reference's linked stream through the linked decode (default path, run walker, pointer pass, run-in decode with random piece and run-in lengths)
    fr_l = o.frame_compress(data, bl, accel, 8, True)
    for env in ({}, {"MI355LZ4_LINKED_RUNS": "100000"}, {"MI355LZ4_LINKED_RUNS": "0"},
                {"MI355LZ4_LINKED_RUNS": "0", "MI355LZ4_LINKED_RUNIN": "1", "MI355LZ4_LINKED_RUNIN_PIECE": str(rng.choice([1, 2, 3, 16])),
                 "MI355LZ4_LINKED_RUNIN_BLOCKS": str(rng.choice([1, 2, 11]))}):
        for k, v in env.items():
            os.environ[k] = v
        out, blen = eng.decompress_batch(fr_l, linked=True)
        for k in env:
            del os.environ[k]
        assert out == data, ("linked", env, bl, nb, lit_max, off_max, mlen_max, accel, cases)
    cases += 1
print("soak ok: %d cases in %.0f s (seed %d)" % (cases, time.time() - t0, seed))
