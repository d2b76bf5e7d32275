"""Development aid: encoder rate + ratio + GPU round-trip check for a few inputs (device-resident, HIP events).
usage: enc_quick.py [kinds=lzsynth,text,random] [nblocks=16384] [block=65536] [accel=1] [linked=0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import torch, streamly_lz4_amd as S
kinds = (sys.argv[1] if len(sys.argv) > 1 else "lzsynth,text,random").split(",")
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
BL = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
accel = int(sys.argv[4]) if len(sys.argv) > 4 else 1
linked = len(sys.argv) > 5 and sys.argv[5] == "1"
dev = torch.device("cuda:0"); eng = S.Engine(0); eng.set_linked_compress(linked)
stride = S.slot_stride(BL, 8)
for kind in kinds:
    src = torch.empty(NB * BL, dtype=torch.uint8, device=dev)
    if kind == "zeros":
        src.zero_()
    else:
        eng.generate(kind, src, BL, NB)
    slots = torch.empty(NB * stride, dtype=torch.uint8, device=dev); flen = torch.empty(NB, dtype=torch.int32, device=dev)
    dense = torch.empty(NB * stride, dtype=torch.uint8, device=dev); doff = torch.empty(NB + 1, dtype=torch.int64, device=dev)
    out = torch.zeros(NB * BL, dtype=torch.uint8, device=dev); res = torch.empty(NB, dtype=torch.int32, device=dev)
    ooff = torch.arange(NB + 1, dtype=torch.int64, device=dev) * BL
    e0, e1 = S.Event(), S.Event()
    best = 1e9
    for _ in range(4):
        eng.record(e0); eng.compress_batch_device(src, NB, BL, slots, stride, flen, accel=accel); eng.record(e1); eng.synchronize()
        best = min(best, eng.elapsed_ms(e0, e1))
    eng.compact_device(slots, stride, flen, NB, dense, NB * stride, doff); eng.synchronize()
    cb = int(doff[-1].item())
    eng.decompress_batch_device(dense, cb, doff, NB, out, ooff, res, linked=linked); eng.synchronize()
    ok = bool((res == BL).all().item()) and torch.equal(out, src)
    print("%-8s %6d x %6d accel %d linked %d: %8.2f GB/s  ratio %.4f  roundtrip %s" % (kind, NB, BL, accel, linked, NB * BL / best / 1e6, NB * BL / cb, "OK" if ok else "MISMATCH"), flush=True)
    if not ok:
        bad = (res != BL).nonzero().flatten()[:8].tolist()
        print("   bad results:", bad, res[bad].tolist() if bad else "", " first diff byte:", int((out != src).nonzero()[0].item()) if not torch.equal(out, src) else -1)
    del src, slots, dense, out
