"""Development aid (CPU only): what the lane-parallel decoder's batches look like on the bench data.

    python scripts/sim/seq_stats.py [lzsynth|text] [n_blocks]

Blocks come from the oracle's generators and its independent-block compressor (the engine's parse is
within 2 % of it in size).  Prints the sequence mix and, for the decoder's batching rule (<= 64 sequences,
tokens inside a 512-byte window, <= 2560 output bytes), the dependency structure of the near matches.
"""
import os
import sys
import collections

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle


def parse(comp):
    """-> list of (tokpos, lit, ml, off, nxt) for every sequence but the last (literal-only) one"""
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
        ip += lit
        if ip >= n:
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
        seqs.append((tp, lit, ml, off, ip))
    return seqs


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "lzsynth"
    nb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    maxseq = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    win = int(sys.argv[4]) if len(sys.argv) > 4 else 512
    o = Oracle()
    BL = 65536
    raw = o.gen(kind, nb, BL)
    H = collections.Counter
    lits, mls, offs, cb = H(), H(), H(), []
    tot_seq = 0
    rounds_h, batch_n = H(), []
    depth_lanes = H()
    inb = 0
    straddle = 0
    lit_only_src = 0
    overlap = 0
    for b in range(nb):
        comp = o.compress_block(raw[b * BL:(b + 1) * BL].tobytes())
        seqs = parse(comp)
        tot_seq += len(seqs)
        op = 0
        outs = []
        for (tp, lit, ml, off, nx) in seqs:
            lits[min(lit, 40)] += 1
            mls[min(ml, 80)] += 1
            offs[min(off >> 8, 16)] += 1
            outs.append(op)
            op += lit + ml
        cb.append(len(comp) / max(len(seqs), 1))
        # batching
        i = 0
        while i < len(seqs):
            w0 = seqs[i][0] & ~15
            j = i
            ostart = outs[i]
            while j < len(seqs) and j - i < maxseq and seqs[j][0] - w0 < win and seqs[j][4] - w0 <= 2 * win and \
                    outs[j] + seqs[j][1] + seqs[j][2] - ostart <= 2560:
                j += 1
            if j == i:
                j = i + 1
            batch_n.append(j - i)
            # dependency depth per sequence inside the batch
            depth = {}
            starts = [outs[k] for k in range(i, j)]
            for k in range(i, j):
                tp, lit, ml, off, nx = seqs[k]
                dpos = outs[k] + lit
                spos = dpos - off
                shi = min(spos + ml, outs[k])       # bytes from outs[k] on are my own literals / myself
                if off < ml:
                    overlap += 1
                d = 0
                if shi > ostart and shi > spos:
                    inb += 1
                    # sequences overlapped
                    lo = max(spos, ostart)
                    ks = [q for q in range(i, k) if outs[q] + seqs[q][1] + seqs[q][2] > lo and outs[q] < shi]
                    only_lit = True
                    for q in ks:
                        qd = outs[q] + seqs[q][1]        # q's match area [qd, qd+ml_q)
                        a, e = max(lo, qd), min(shi, qd + seqs[q][2])
                        if e > a:
                            only_lit = False
                            d = max(d, depth[q] + 1)
                    if only_lit:
                        lit_only_src += 1
                    if len(ks) > 1:
                        straddle += 1
                depth[k] = d
                depth_lanes[d] += 1
            rounds_h[max(depth.values()) + 1] += 1
            i = j
    print("== %s: %d blocks, %.1f seq/block, %.2f compressed bytes/seq, %.1f out bytes/seq" %
          (kind, nb, tot_seq / nb, float(np.mean(cb)), BL * nb / tot_seq))

    def show(name, h, n=tot_seq):
        print("  %s: " % name + " ".join("%d:%.1f%%" % (k, 100.0 * v / n) for k, v in sorted(h.items())))
    show("lit", lits)
    show("ml", mls)
    show("off>>8", offs)
    nbat = len(batch_n)
    print("  batches/block %.1f, seq/batch %.1f" % (nbat / nb, float(np.mean(batch_n))))
    show("rounds/batch", rounds_h, nbat)
    show("match depth (lanes)", depth_lanes)
    print("  in-batch source %.1f%% of seqs; of those: source only literals %.1f%%, touches >1 seq %.1f%%; offset<ml %.2f%%" %
          (100.0 * inb / tot_seq, 100.0 * lit_only_src / max(inb, 1), 100.0 * straddle / max(inb, 1), 100.0 * overlap / tot_seq))


main()
