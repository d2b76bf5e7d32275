"""Development aid: run a few decode launches (for rocprofv3).  python scripts/prof_decode.py [kind] [nblocks] [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import torch, streamly_lz4_amd as S
kind = sys.argv[1] if len(sys.argv) > 1 else "lzsynth"
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
BL = 65536
dev = torch.device("cuda:0"); eng = S.Engine(0)
eng.set_decoder(int(os.environ.get("DEC_VARIANT", "0")))
src = torch.empty(NB * BL, dtype=torch.uint8, device=dev); eng.generate(kind, src, BL, NB)
stride = S.slot_stride(BL, 8)
slots = torch.empty(NB * stride, dtype=torch.uint8, device=dev); flen = torch.empty(NB, dtype=torch.int32, device=dev)
dense = torch.empty(NB * stride, dtype=torch.uint8, device=dev); doff = torch.empty(NB + 1, dtype=torch.int64, device=dev)
ooff = torch.arange(NB + 1, dtype=torch.int64, device=dev) * BL
out = torch.empty(NB * BL, dtype=torch.uint8, device=dev); res = torch.empty(NB, dtype=torch.int32, device=dev)
eng.compress_batch_device(src, NB, BL, slots, stride, flen)
eng.compact_device(slots, stride, flen, NB, dense, NB * stride, doff); eng.synchronize()
for _ in range(reps):
    eng.decompress_batch_device(dense, NB * stride, doff, NB, out, ooff, res)
eng.synchronize()
print("ok", bool((res == BL).all().item()), "U", NB * BL, "C", int(doff[-1].item()))
