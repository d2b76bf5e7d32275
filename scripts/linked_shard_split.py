"""Development aid: how a linked decode of one RANGE splits into the half that needs nothing from the left neighbour
(mi355lz4_decompress_linked_begin) and the half that waits for the seam (_end).  One GPU, reference-written text stream."""
import os, struct, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import numpy as np, torch, streamly_lz4_amd as S
from oracle.oracle import Oracle, Reference, have_reference
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
BL = 65536
O = Reference() if have_reference() else Oracle()
eng = S.Engine(0)
src = torch.empty(NB * BL, dtype=torch.uint8, device="cuda"); eng.generate("text", src, BL, NB); eng.synchronize()
raw = src.cpu().numpy().tobytes()
framed = O.frame_compress(raw, BL, 1, 8, True)
offs, pos = [], 0
while pos < len(framed):
    offs.append(pos); pos += 8 + struct.unpack_from("<i", framed, pos)[0]
offs.append(pos)
fr = torch.from_numpy(np.frombuffer(framed, dtype=np.uint8).copy()).cuda()
boff = torch.tensor(offs, dtype=torch.int64).cuda()
ooff = (torch.arange(NB + 1, dtype=torch.int64) * BL).cuda()
out = torch.zeros(NB * BL, dtype=torch.uint8, device="cuda"); res = torch.zeros(NB, dtype=torch.int32, device="cuda")
e = [S.Event() for _ in range(3)]
best = None
for _ in range(4):
    eng.record(e[0]); eng.decompress_linked_begin(fr, len(framed), boff, NB, out, ooff, res, 0)
    eng.record(e[1]); eng.decompress_linked_end(); eng.record(e[2]); eng.synchronize()
    t = (eng.elapsed_ms(e[0], e[1]), eng.elapsed_ms(e[1], e[2]))
    best = t if best is None or sum(t) < sum(best) else best
ok = torch.equal(out, src)
# ... and with the range's last block made final ahead of the others (what the right neighbour waits for): wall clock,
# the call waits for its result
import time
last_ms, last_ok = None, None
for _ in range(4):
    out.zero_()
    eng.decompress_linked_begin(fr, len(framed), boff, NB, out, ooff, res, 0); eng.synchronize()
    t0 = time.perf_counter(); early = eng.decompress_linked_end_last(); dt = (time.perf_counter() - t0) * 1e3
    last_ok = bool(early) and torch.equal(out[(NB - 1) * BL:], src[(NB - 1) * BL:])
    eng.decompress_linked_end(); eng.synchronize()
    last_ms = dt if last_ms is None else min(last_ms, dt)
ok = ok and torch.equal(out, src)
e0, e1 = S.Event(), S.Event()
eng.record(e0); eng.decompress_batch_device(fr, len(framed), boff, NB, out, ooff, res, linked=True); eng.record(e1); eng.synchronize()
one = eng.elapsed_ms(e0, e1)
print({"blocks": NB, "MiB": NB * BL >> 20, "begin_ms": round(best[0], 3), "end_ms": round(best[1], 3), "one_call_ms": round(one, 3),
       "GBps_one_call": round(NB * BL / one / 1e6, 1), "serial_fraction": round(best[1] / sum(best), 3),
       "end_last_ms": round(last_ms, 3), "end_last_final": last_ok, "verified": bool(ok)})
