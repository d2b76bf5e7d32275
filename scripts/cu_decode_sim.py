#!/usr/bin/env python3
"""CPU statistics behind the workgroup-per-block decoder (decode_cu.hpp), on reference-written blocks.

(1) Depth of the sequence dependency graph when literals are free: a match is ready once every sequence whose MATCH
    bytes its source overlaps is complete (literal bytes are all written before the first match).  The number of
    rounds a round-synchronous copy needs is the largest depth; the histogram says how thin the tail is.
(2) Self-synchronisation of the token chain: start parsing at an arbitrary byte LOOKBACK bytes in front of a chunk
    boundary -- how often is the first token at or behind the boundary a TRUE token?  (The parse guesses every
    chunk's entry this way and verifies the guesses against each other.)

usage: cu_decode_sim.py [lzsynth|text|pysrc] [blocks] [blockLen]
"""
import glob
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle  # noqa: E402


def parse(comp):
    """[(tokenPos, litStart, lit, off, ml, next)] of a valid block; the last sequence has ml = 0."""
    seqs = []
    ip, n = 0, len(comp)
    while ip < n:
        tp = ip
        t = comp[ip]; ip += 1
        lit = t >> 4
        if lit == 15:
            while True:
                b = comp[ip]; ip += 1
                lit += b
                if b != 255:
                    break
        ls = ip
        ip += lit
        if ip >= n:
            seqs.append((tp, ls, lit, 0, 0, ip))
            break
        off = comp[ip] | (comp[ip + 1] << 8); ip += 2
        ml = t & 15
        if ml == 15:
            while True:
                b = comp[ip]; ip += 1
                ml += b
                if b != 255:
                    break
        ml += 4
        seqs.append((tp, ls, lit, off, ml, ip))
    return seqs


def succ_spec(comp, p, n):
    """successor of byte position p read as a token, the way the speculative walk does (one extension byte at most);
    None = unknown"""
    if p + 1 >= n:
        return None
    t = comp[p]
    lit = t >> 4
    q = p + 1
    if lit == 15:
        b = comp[q]; q += 1
        if b == 255:
            return None
        lit += b
    q += lit + 2
    if q >= n:
        return None
    if (t & 15) == 15:
        if comp[q] == 255:
            return None
        q += 1
    return q


def depth_stats(seqs, out_len):
    owner = np.full(out_len + 1, -1, dtype=np.int32)     # match index that wrote a byte, -1 = literal
    depth = np.zeros(len(seqs), dtype=np.int32)
    op = 0
    for i, (tp, ls, lit, off, ml, nx) in enumerate(seqs):
        op += lit
        if ml == 0:
            break
        s = op - off
        hi = min(s + ml, op)
        src = owner[s:hi]
        src = src[src >= 0]
        d = 1
        if src.size:
            d = 1 + int(depth[np.unique(src)].max())
        depth[i] = d
        owner[op:op + ml] = i
        op += ml
    return depth


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "lzsynth"
    nblk = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    blen = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
    o = Oracle()
    if kind == "pysrc":
        data = b"".join(open(f, "rb").read() for f in sorted(glob.glob("/usr/lib/python3.10/*.py")))
        raw = np.frombuffer(data[: nblk * blen], dtype=np.uint8)
    else:
        raw = o.gen(kind, nblk, blen)
    S, LOOK = 32, 256
    maxd, hist = [], np.zeros(4096, dtype=np.int64)
    miss = tot = 0
    nseq = []
    for b in range(nblk):
        blk = raw[b * blen:(b + 1) * blen].tobytes()
        comp = o.compress_block(blk, 1)
        seqs = parse(comp)
        nseq.append(len(seqs))
        d = depth_stats(seqs, len(blk))
        maxd.append(int(d.max()))
        hist[: d.max() + 1] += np.bincount(d, minlength=d.max() + 1)[: d.max() + 1]
        true = set(s[0] for s in seqs)
        tok = sorted(true)
        n = len(comp)
        ti = 0
        for c in range(1, n // S):
            bound = c * S
            p = max(0, bound - LOOK)
            while p is not None and p < bound:
                p = succ_spec(comp, p, n)
            while ti < len(tok) and tok[ti] < bound:
                ti += 1
            if ti >= len(tok):
                break
            tot += 1
            if p != tok[ti]:
                miss += 1
    hist = hist[: max(maxd) + 1]
    cum = np.cumsum(hist[1:]) / max(1, hist[1:].sum())
    print(f"{kind}: {nblk} blocks of {blen}: sequences/block {np.mean(nseq):.0f}; depth max per block {maxd}")
    for q in (0.5, 0.9, 0.99, 0.999):
        print(f"  {q * 100:.1f}% of matches at depth <= {int(np.searchsorted(cum, q)) + 1}")
    print(f"  entry guesses (chunks of {S}, look-back {LOOK}): {miss} of {tot} wrong ({100.0 * miss / max(1, tot):.2f}%)")


if __name__ == "__main__":
    main()
