#!/usr/bin/env python3
"""Development aid (CPU only): how many blocks a decode that starts WITHOUT its dictionary needs until a block comes
out exact -- the run-in of the linked decode's pieces (kernels.hip, "RUN-IN DECODE").  A reference-written linked stream
(oracle compressor == reference's, byte for byte); for every block as a starting point, the bytes that derive from the
missing dictionary are followed block by block (byte-exact) until a block has none.
usage: runin_sim.py [kind=text|lzsynth|pysrc|periodic] [blocks=120] [block_len=65536]"""
import glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle.oracle import Oracle

kind = sys.argv[1] if len(sys.argv) > 1 else "text"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 120
bl = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
O = Oracle()
if kind == "pysrc":
    data = b"".join(open(f, "rb").read() for f in sorted(glob.glob("/usr/lib/python3.10/*.py")))
    nb = min(nb, len(data) // bl); data = data[: nb * bl]
elif kind == "periodic":
    pat = O.gen("text", 1, 3001, first_block=7).tobytes()
    data = (pat * (nb * bl // len(pat) + 1))[: nb * bl]
else:
    data = O.gen(kind, nb, bl, first_block=7).tobytes()
fr = O.frame_compress(data, bl, 1, 8, True)

def parse(block):
    seqs = []; ip = 0; n = len(block)
    while ip < n:
        t = block[ip]; ip += 1
        lit = t >> 4
        if lit == 15:
            while True:
                b = block[ip]; ip += 1; lit += b
                if b != 255: break
        ip += lit
        if ip >= n:
            seqs.append((lit, 0, 0)); break
        off = block[ip] | (block[ip + 1] << 8); ip += 2
        ml = (t & 15) + 4
        if (t & 15) == 15:
            while True:
                b = block[ip]; ip += 1; ml += b
                if b != 255: break
        seqs.append((lit, off, ml))
    return seqs

blocks = []; pos = 0
for _ in range(nb):
    c = int.from_bytes(fr[pos:pos + 4], "little"); blocks.append(parse(fr[pos + 8:pos + 8 + c])); pos += 8 + c

def step(prev, b):
    """prev: which bytes of the dictionary block are not exact; returns the same for this block"""
    buf = np.zeros(2 * bl, dtype=bool); buf[:bl] = prev
    op = bl
    for lit, off, ml in b:
        op += lit
        if ml == 0: break
        src = op - off
        if off >= ml:
            buf[op:op + ml] = buf[src:src + ml]
        else:
            pat = buf[src:op]
            if pat.any():
                buf[op:op + ml] = np.tile(pat, (ml + off - 1) // off)[:ml]
        op += ml
    return buf[bl:]

MAXD = min(32, nb // 2)
hist = {}
for s in range(1, nb - MAXD):
    t = np.ones(bl, dtype=bool); d = 0
    while d < MAXD:
        t = step(t, blocks[s + d]); d += 1
        if not t.any(): break
    else:
        d = MAXD + 1
    hist[d] = hist.get(d, 0) + 1
tot = sum(hist.values()); cum = 0
print("%s, %d blocks of %d (ratio %.3f): the first exact block of a decode that starts without its dictionary" % (kind, nb, bl, len(data) / len(fr)))
for d in sorted(hist):
    cum += hist[d]
    print("  %s: %5.1f %% of the starts, cumulative %5.1f %%" % ("block %2d" % d if d <= MAXD else "none in %d" % MAXD, 100.0 * hist[d] / tot, 100.0 * cum / tot))
