"""Development aid: compress latency of small calls (device-resident), segment mode on/off via MI355LZ4_SEG."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import torch, streamly_lz4_amd as S
dev = torch.device("cuda:0"); eng = S.Engine(0)
for kind in ("text", "lzsynth"):
    for NB, BL in ((16, 65536), (160, 65536), (1024, 65536), (16, 655360), (4, 4 << 20)):
        src = torch.empty(NB * BL, dtype=torch.uint8, device=dev); eng.generate(kind, src, BL, NB)
        stride = S.slot_stride(BL, 8)
        slots = torch.empty(NB * stride, dtype=torch.uint8, device=dev); flen = torch.empty(NB, dtype=torch.int32, device=dev)
        dense = torch.empty(NB * stride, dtype=torch.uint8, device=dev); doff = torch.empty(NB + 1, dtype=torch.int64, device=dev)
        out = torch.zeros(NB * BL, dtype=torch.uint8, device=dev); res = torch.empty(NB, dtype=torch.int32, device=dev)
        ooff = torch.arange(NB + 1, dtype=torch.int64, device=dev) * BL
        e0, e1 = S.Event(), S.Event(); best = 1e9
        for _ in range(5):
            eng.record(e0); eng.compress_batch_device(src, NB, BL, slots, stride, flen); eng.record(e1); eng.synchronize()
            best = min(best, eng.elapsed_ms(e0, e1))
        eng.compact_device(slots, stride, flen, NB, dense, NB * stride, doff); eng.synchronize()
        cb = int(doff[-1].item())
        eng.decompress_batch_device(dense, cb, doff, NB, out, ooff, res); eng.synchronize()
        ok = bool((res == BL).all().item()) and torch.equal(out, src)
        print("%-8s %5d x %8d: %8.3f ms  %7.2f GB/s  ratio %.4f  %s" % (kind, NB, BL, best, NB * BL / best / 1e6, NB * BL / cb, "OK" if ok else "MISMATCH"), flush=True)
