"""Development aid: how often the token that follows a token starts in the same 8-byte slot of the compressed stream
(what resolving a lane's in-slot hops in registers before the first squaring round of k_decode_par could save), and
how many compressed bytes a sequence takes.  CPU only: blocks written by the oracle's compressor."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.oracle import Oracle, Reference, have_reference
O = Oracle()
BL = 65536
for kind in ("lzsynth", "text"):
    same = total = cbytes = 0
    for b in range(8):
        raw = O.gen(kind, 1, BL, first_block=b).tobytes()
        c = O.compress_block(raw, 1)
        ip, n, toks = 0, len(c), []
        while ip < n:
            toks.append(ip)
            t = c[ip]; ip += 1
            lit = t >> 4
            if lit == 15:
                while True:
                    x = c[ip]; ip += 1; lit += x
                    if x != 255: break
            ip += lit
            if ip >= n: break
            ip += 2
            ml = t & 15
            if ml == 15:
                while True:
                    x = c[ip]; ip += 1
                    if x != 255: break
        for a, nx in zip(toks, toks[1:]):
            total += 1
            cbytes += nx - a
            # slots are 8-byte aligned pieces of the window; the window base is 16-byte aligned, take the block start as base
            if (a >> 3) == (nx >> 3): same += 1
    print({"kind": kind, "sequences": total, "compressed_bytes_per_sequence": round(cbytes / total, 2),
           "next_token_in_same_8_byte_slot": round(same / total, 4)})
