#!/usr/bin/env python3
"""Development aid (CPU only): what the tolerant pass of the linked decode would defer at different taint granules, on a
reference-written linked stream (oracle compressor == reference's, byte for byte).

For every block behind the first, every match is either copyable without the dictionary or DEFERRED: its source starts
in front of the block, or touches output that a deferred match wrote (tracked per granule of G bytes).  Prints, per
granule: deferred matches and bytes per block, and -- for the exact case G = 1 -- the histogram of chain depths (how
many deferred hops until a byte's root) in matches and in blocks.
usage: tol_taint_sim.py [kind=text] [blocks=32] [block_len=65536]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle.oracle import Oracle

kind = sys.argv[1] if len(sys.argv) > 1 else "text"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 32
bl = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
O = Oracle()
data = O.gen(kind, nb, bl, first_block=7).tobytes()
fr = O.frame_compress(data, bl, 1, 8, True)

def parse(block):
    """sequences of one LZ4 block: (literal length, offset, match length), the last one with offset 0"""
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

print("%s: %d blocks of %d bytes, %.0f sequences per block" % (kind, nb, bl, sum(len(b) for b in blocks) / nb))
for G in (16, 4, 1):
    dm = db = tot_m = 0
    for b in blocks[1:]:
        taint = np.zeros((bl + G - 1) // G + 1, dtype=bool)
        op = 0
        for lit, off, ml in b:
            op += lit
            if ml == 0: break
            tot_m += 1
            src = op - off
            defer = src < 0
            if not defer:
                hi = min(src + ml, op)
                defer = hi > src and taint[src // G:(hi - 1) // G + 1].any()
            if defer:
                dm += 1; db += ml
                taint[op // G:(op + ml - 1) // G + 1] = True
            op += ml
    print("  granule %2d: deferred %.0f of %.0f matches per block (%.1f %%), %.0f bytes per block (%.1f %% of the output)"
          % (G, dm / (nb - 1), tot_m / (nb - 1), 100.0 * dm / tot_m, db / (nb - 1), 100.0 * db / ((nb - 1) * bl)))

# exact chains: per byte, depth in deferred hops and in blocks (G = 1).  dep[b][i] = (hops, blocks back) of byte i of block b
prev_h = np.zeros(bl, dtype=np.int32); prev_k = np.zeros(bl, dtype=np.int32)    # block 0: all roots
hist_h = np.zeros(4096, dtype=np.int64); hist_k = np.zeros(nb + 1, dtype=np.int64)
for bi, b in enumerate(blocks[1:], 1):
    h = np.zeros(bl, dtype=np.int32); k = np.zeros(bl, dtype=np.int32); dep = np.zeros(bl, dtype=bool)
    op = 0
    for lit, off, ml in b:
        op += lit
        if ml == 0: break
        src = op - off
        for j in range(ml):                       # byte by byte: overlapping matches read what they wrote
            s = src + j
            if s < 0:
                dep[op + j] = True; h[op + j] = prev_h[bl + s] + 1; k[op + j] = prev_k[bl + s] + 1
            elif dep[s]:
                dep[op + j] = True; h[op + j] = h[s] + 1; k[op + j] = k[s]
        op += ml
    d = dep.nonzero()[0]
    np.add.at(hist_h, np.minimum(h[d], 4095), 1); np.add.at(hist_k, k[d], 1)
    prev_h = np.where(dep, h, 0); prev_k = np.where(dep, k, 0)
tot = hist_h.sum()
cum = np.cumsum(hist_h) / max(tot, 1)
print("  exact chains (bytes that need the dictionary: %.1f %% of the output): hops 1: %.1f %%, <= 2: %.1f %%, <= 4: %.1f %%, <= 8: %.1f %%, <= 16: %.1f %%, <= 64: %.1f %%, max %d"
      % (100.0 * tot / ((nb - 1) * bl), 100 * cum[1], 100 * cum[2], 100 * cum[4], 100 * cum[8], 100 * cum[16], 100 * cum[64], int(hist_h.nonzero()[0].max())))
ck = np.cumsum(hist_k) / max(tot, 1)
print("  blocks back to the root: 1: %.1f %%, <= 2: %.1f %%, <= 4: %.1f %%, <= 8: %.1f %%, max %d" % (100 * ck[1], 100 * ck[2], 100 * ck[4], 100 * ck[min(8, nb)], int(hist_k.nonzero()[0].max())))

# ---- runs of B consecutive blocks decoded by one wave, each block with the block before it (in the run) as dictionary:
# only the run's first block lacks its dictionary.  Deferred bytes per block position in the run, match-level taint at
# granule G (what TolCtx does today) and byte-exact (a match is split at the bytes that depend on the missing dictionary).
print("runs of B blocks, deferred bytes (%% of a block) by position in the run:")
for G in (16, 1, 0):                      # 0 = byte-exact
    B = 8
    acc = np.zeros(B); cnt = np.zeros(B)
    for r0 in range(1, nb - B + 1, B):
        prev = np.ones((bl + max(G, 1) - 1) // max(G, 1) + 1, dtype=bool) if G else np.ones(bl, dtype=bool)    # the missing dictionary: all tainted
        for j in range(B):
            b = blocks[r0 + j]
            if G:
                taint = np.zeros((bl + G - 1) // G + 1, dtype=bool); db = 0; op = 0
                for lit, off, ml in b:
                    op += lit
                    if ml == 0: break
                    src = op - off
                    defer = False
                    if src < 0:
                        lo = bl + src; hi = min(bl, lo + ml)
                        defer = prev[lo // G:(hi - 1) // G + 1].any()
                        if not defer and src + ml > 0:
                            defer = taint[0:(min(src + ml, op) - 1) // G + 1].any()
                    else:
                        hi = min(src + ml, op)
                        defer = hi > src and taint[src // G:(hi - 1) // G + 1].any()
                    if defer:
                        db += ml; taint[op // G:(op + ml - 1) // G + 1] = True
                    op += ml
                prev = taint
            else:
                dep = np.zeros(bl, dtype=bool); op = 0
                for lit, off, ml in b:
                    op += lit
                    if ml == 0: break
                    src = op - off
                    for q in range(ml):
                        s = src + q
                        dep[op + q] = prev[bl + s] if s < 0 else dep[s]
                    op += ml
                db = int(dep.sum()); prev = dep
            acc[j] += db; cnt[j] += 1
    print("  %s: " % ("granule %2d, whole matches" % G if G else "byte-exact             ") + "  ".join("%5.1f" % (100.0 * a / c / bl) for a, c in zip(acc, cnt)))
