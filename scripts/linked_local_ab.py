"""Development aid: one reference-written linked text stream, second pass by the pointer pass (MI355LZ4_LINKED_LOCAL=0)
against local resolve + chase (linked_ptr.hpp).  python scripts/linked_local_ab.py [blocks] [kind]"""
import os, struct, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import numpy as np, torch, streamly_lz4_amd as S
from oracle.oracle import Oracle, Reference, have_reference
O = Reference() if have_reference() else Oracle()
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
kind = sys.argv[2] if len(sys.argv) > 2 else "text"
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["0", "1"]
eng = S.Engine(0); BL = 65536; dev = "cuda"
def timed(fn, n=5):
    e0, e1 = S.Event(), S.Event(); best = 1e9
    for _ in range(n):
        eng.record(e0); fn(); eng.record(e1); eng.synchronize(); best = min(best, eng.elapsed_ms(e0, e1))
    return best
src = torch.empty(NB * BL, dtype=torch.uint8, device=dev); eng.generate(kind, src, BL, NB); eng.synchronize()
raw = src.cpu().numpy().tobytes()
framed = O.frame_compress(raw, BL, 1, 8, True)
offs, pos = [], 0
while pos < len(framed):
    offs.append(pos); pos += 8 + struct.unpack_from("<i", framed, pos)[0]
offs.append(pos)
fr = torch.from_numpy(np.frombuffer(framed, dtype=np.uint8).copy()).cuda(); boff = torch.tensor(offs, dtype=torch.int64).cuda()
out = torch.zeros(NB * BL, dtype=torch.uint8, device=dev); res = torch.zeros(NB, dtype=torch.int32, device=dev)
ooff = torch.arange(NB + 1, dtype=torch.int64, device=dev) * BL
for mode in modes:
    os.environ["MI355LZ4_LINKED_LOCAL"] = mode
    out.zero_()
    ms = timed(lambda: eng.decompress_batch_device(fr, len(framed), boff, NB, out, ooff, res, linked=True))
    ok = bool(torch.equal(out, src)) and bool((res == BL).all().item())
    print({"blocks": NB, "kind": kind, "local": mode, "ms": round(ms, 3), "GBps": round(NB * BL / ms / 1e6, 1), "ok": ok}, flush=True)
