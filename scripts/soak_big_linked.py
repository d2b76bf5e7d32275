"""Development aid (GPU box): randomized differential soak of the big-linked-blocks path (api.cpp path 6: guessed dictionaries)
against the pointer pass (MI355LZ4_LINKED_BIG=0) and the input: block sizes from 512 KiB to 3 MiB, ragged last blocks, text /
lzsynth / mixed / periodic data, streams written by the reference or by the engine's linked compressor, clean and corrupted.
    python scripts/soak_big_linked.py [seconds] [seed]"""
import ctypes as C, os, random, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import numpy as np, torch, streamly_lz4_amd as S
from oracle.oracle import Oracle
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
O = Oracle(); eng = S.Engine(0); dev = torch.device("cuda:0")
os.environ.pop("MI355LZ4_LINKED_BIG", None)


def make(kind, n):
    nb = (n + 65535) // 65536
    if kind == "text": return O.gen("text", nb, 65536, first_block=rng.randrange(1000)).tobytes()[:n]
    if kind == "lzsynth": return O.gen("lzsynth", nb, 65536, first_block=rng.randrange(1000)).tobytes()[:n]
    if kind == "mixed":
        parts = []
        while sum(map(len, parts)) < n:
            parts.append(make(rng.choice(("text", "lzsynth", "random", "zeros")), rng.randrange(20000, 400000)))
        return b"".join(parts)[:n]
    if kind == "random": return rng.randbytes(n)
    if kind == "zeros": return bytes(n)
    pat = rng.randbytes(rng.choice((700, 5000, 61000, 65000, 70000)))      # periodic: blocks that lean on their dictionary for long
    return (pat * (n // len(pat) + 1))[:n]


def call(fr, nblk, bl, n):
    offs, pos = [], 0
    for _ in range(nblk):
        offs.append(pos); pos += 8 + int.from_bytes(fr[pos:pos + 4], "little")
    buf = torch.frombuffer(bytearray(fr), dtype=torch.uint8).to(dev)
    boff = torch.tensor(offs + [pos], dtype=torch.int64, device=dev)
    ooff = torch.arange(0, (nblk + 1) * bl, bl, dtype=torch.int64, device=dev)
    out = torch.zeros(nblk * bl, dtype=torch.uint8, device=dev); res = torch.zeros(nblk, dtype=torch.int32, device=dev)
    eng.decompress_batch_device(buf, len(fr), boff, nblk, out, ooff, res, linked=True); eng.synchronize()
    st = (C.c_int * 5)(); S.lib.mi355lz4_debug_runin_state(eng.ctx, st, None)
    return out.cpu().numpy().tobytes()[:n], res.cpu().tolist(), st[4], st[3]


t0 = time.time(); cases = 0; paths = {}
while time.time() - t0 < budget:
    bl = rng.choice((512 << 10, 768 << 10, 1 << 20, 1536 << 10, 2 << 20, 3 << 20))
    nblk = rng.randrange(2, 14)
    n = nblk * bl - rng.choice((0, 0, 1, 777, 65536, bl // 2))
    kind = rng.choice(("text", "text", "lzsynth", "mixed", "mixed", "periodic"))
    raw = make(kind, n)
    if rng.random() < 0.3:
        e2 = S.Engine(0); e2.set_linked_compress(True)
        fr, _ = e2.compress_batch([raw[i:i + bl] for i in range(0, n, bl)]); e2.close()
    else:
        fr = O.frame_compress(raw, bl, 1, 8, True)
    corrupt = rng.random() < 0.35
    if corrupt:
        b = bytearray(fr)
        for _ in range(rng.choice((1, 1, 3))):
            p = rng.randrange(16, len(b) - 64); b[p:p + rng.choice((1, 4, 40))] = bytes([rng.choice((0, 255, rng.randrange(256)))]) * rng.choice((1, 4, 40))
        # (headers stay plausible: a broken size field is the header checks' business, tested elsewhere)
        pos = 0
        for k in range(nblk):
            b[pos:pos + 8] = fr[pos:pos + 8]; pos += 8 + int.from_bytes(fr[pos:pos + 4], "little")
        fr = bytes(b)
    o1, r1, p1, passes = call(fr, nblk, bl, n)
    os.environ["MI355LZ4_LINKED_BIG"] = "0"
    o0, r0, p0, _ = call(fr, nblk, bl, n)
    os.environ.pop("MI355LZ4_LINKED_BIG")
    paths[p1] = paths.get(p1, 0) + 1
    if r1 != r0:
        sys.exit("results differ: case %d %s bl %d nblk %d corrupt %s path %d: %s" % (cases, kind, bl, nblk, corrupt, p1, [(i, a, c) for i, (a, c) in enumerate(zip(r1, r0)) if a != c][:6]))
    good = 0
    while good < nblk and r0[good] > 0: good += 1
    end = min(n, good * bl)
    if o1[:end] != o0[:end] or (not corrupt and o1 != raw):
        a = np.frombuffer(o1[:end], np.uint8); c = np.frombuffer(o0[:end], np.uint8)
        sys.exit("bytes differ: case %d %s bl %d nblk %d corrupt %s path %d at %s" % (cases, kind, bl, nblk, corrupt, p1, np.nonzero(a != c)[0][:4]))
    cases += 1
print("big-linked soak ok: %d cases in %.0f s, paths %s" % (cases, time.time() - t0, paths))
