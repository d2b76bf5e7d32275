"""Development aid: a few small device-resident decodes (160 blocks of 64 KiB, the reference's 10 MiB protocol) with the
decoder the engine picks for them -- one workgroup per block, csrc/decode_cu.hpp -- for rocprofv3 (--kernel-trace --stats, --pmc).
    python scripts/prof_cu.py [kind] [blocks] [reps] [blockLen]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import torch, streamly_lz4_amd as S
kind = sys.argv[1] if len(sys.argv) > 1 else "lzsynth"
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 160
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
BL = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
dev = torch.device("cuda:0")
eng = S.Engine(0)
# (big blocks: the generators make 64 KiB blocks of their own seed, a big block is a run of them)
src = torch.empty(NB * BL, dtype=torch.uint8, device=dev); eng.generate(kind, src, 65536, NB * BL // 65536)
stride = S.slot_stride(BL, 8)
slots = torch.empty(NB * stride, dtype=torch.uint8, device=dev); flen = torch.empty(NB, dtype=torch.int32, device=dev)
dense = torch.empty(NB * stride, dtype=torch.uint8, device=dev); doff = torch.empty(NB + 1, dtype=torch.int64, device=dev)
ooff = torch.arange(NB + 1, dtype=torch.int64, device=dev) * BL
out = torch.empty(NB * BL, dtype=torch.uint8, device=dev); res = torch.empty(NB, dtype=torch.int32, device=dev)
eng.compress_batch_device(src, NB, BL, slots, stride, flen)
eng.compact_device(slots, stride, flen, NB, dense, NB * stride, doff); eng.synchronize()
C = int(doff[-1].item())
for _ in range(reps):
    eng.decompress_batch_device(dense, C, doff, NB, out, ooff, res)
eng.synchronize()
print("ok", bool((res == BL).all().item()) and torch.equal(out, src), kind, NB, "blocks of", BL, "reps", reps, "compressed", C)
