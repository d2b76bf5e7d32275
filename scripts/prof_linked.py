"""Development aid: a few linked decodes of ONE reference-written text stream (for rocprofv3 --kernel-trace --stats: the
second pass kernel by kernel).  python scripts/prof_linked.py [blocks] [reps]"""
import os, struct, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import numpy as np, torch, streamly_lz4_amd as S
from oracle.oracle import Oracle, Reference, have_reference
O = Reference() if have_reference() else Oracle()
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
BL = 65536
eng = S.Engine(0)
src = torch.empty(NB * BL, dtype=torch.uint8, device="cuda"); eng.generate("text", src, BL, NB); eng.synchronize()
framed = O.frame_compress(src.cpu().numpy().tobytes(), BL, 1, 8, True)
offs, pos = [], 0
while pos < len(framed):
    offs.append(pos); pos += 8 + struct.unpack_from("<i", framed, pos)[0]
offs.append(pos)
fr = torch.from_numpy(np.frombuffer(framed, dtype=np.uint8).copy()).cuda(); boff = torch.tensor(offs, dtype=torch.int64).cuda()
out = torch.zeros(NB * BL, dtype=torch.uint8, device="cuda"); res = torch.zeros(NB, dtype=torch.int32, device="cuda")
ooff = torch.arange(NB + 1, dtype=torch.int64, device="cuda") * BL
for _ in range(reps):
    eng.decompress_batch_device(fr, len(framed), boff, NB, out, ooff, res, linked=True)
eng.synchronize()
print("ok", bool(torch.equal(out, src)), "blocks", NB, "reps", reps)
