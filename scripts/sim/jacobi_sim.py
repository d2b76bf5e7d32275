"""Development aid: simulate the segment-parallel (Jacobi) token parse on real compressed blocks.
   python scripts/sim/jacobi_sim.py  -> iterations / steps per chunk for S in {32, 64, 128}"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle.oracle import Oracle

O = Oracle()

def nxt_of(b, p, lim):
    """successor of a token at p (single-byte extensions assumed), or None if it does not fit below lim"""
    if p + 1 >= len(b): return None
    t = b[p]; lit = t >> 4
    q = p + 1
    if lit == 15:
        lit += b[q]; q += 1
    q += lit + 2
    if (t & 15) == 15: q += 1
    return q if q <= lim else None

def sim_chunk(b, start, W, S, NT):
    """b: compressed block bytes; window base = start & ~15; returns (iterations, total steps, n tokens, next ip)"""
    base = start & ~15
    lim = min(len(b) - 32, base + W)
    L = 64
    INF = 1 << 30
    entry = [base + S * l for l in range(L)]
    entry[0] = start
    iters = 0; steps_total = 0
    while True:
        iters += 1
        exits = []; cnts = []; maxsteps = 0
        for l in range(L):
            p = entry[l]; c = 0
            segEnd = base + S * (l + 1)
            while p < segEnd:
                n = nxt_of(b, p, lim)
                if n is None:
                    p = INF; break
                p = n; c += 1
            exits.append(p); cnts.append(c); maxsteps = max(maxsteps, c)
        steps_total += maxsteps + 1
        new = [start] + exits[:-1]
        if new == entry:
            break
        entry = new
    n = sum(cnts)
    # true chain check
    p = start; k = 0
    while True:
        q = nxt_of(b, p, lim)
        if q is None: break
        p = q; k += 1
    assert k == n, (k, n)
    return iters, steps_total + maxsteps + 1, min(n, NT), p, maxsteps

for kind in ("lzsynth", "text"):
    raw = O.gen(kind, 4, 65536).tobytes()
    for S in (16, 32, 64, 128):
        W = 64 * S
        NT = 1 << 20
        its = []; steps = []; toks = []; ms = []
        for bi in range(4):
            comp = np.frombuffer(O.compress_block(raw[bi * 65536:(bi + 1) * 65536], 1), dtype=np.uint8).tolist()
            ip = 0
            while len(comp) - ip >= 64:
                it, st, n, nip, m = sim_chunk(comp, ip, W, S, NT)
                if n == 0: break
                its.append(it); steps.append(st); toks.append(n); ms.append(m)
                ip = nip
        its = np.array(its); steps = np.array(steps); toks = np.array(toks)
        print("%s S=%d W=%d: chunks/blk %.1f  tokens/chunk %.0f  iters mean %.2f max %d  maxsteps/iter %.1f  SIMT steps (incl final pass)/chunk %.1f -> steps per 64 tokens %.2f"
              % (kind, S, W, len(its) / 4, toks.mean(), its.mean(), its.max(), np.mean(ms), steps.mean(), 64 * steps.sum() / toks.sum()))
