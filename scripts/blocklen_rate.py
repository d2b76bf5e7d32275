"""Development aid: decode rate against block length (same total bytes): what a block's start and its sequential tail cost.
    python scripts/blocklen_rate.py [kind] [total MiB]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import torch, streamly_lz4_amd as S
dev = torch.device("cuda:0"); eng = S.Engine(0)
kind = sys.argv[1] if len(sys.argv) > 1 else "lzsynth"
total = (int(sys.argv[2]) if len(sys.argv) > 2 else 2048) << 20
for BL in (16384, 32768, 65536, 131072, 262144):
    NB = total // BL
    src = torch.empty(NB * BL, dtype=torch.uint8, device=dev); eng.generate(kind, src, BL, NB)
    stride = S.slot_stride(BL, 8)
    slots = torch.empty(NB * stride, dtype=torch.uint8, device=dev); flen = torch.empty(NB, dtype=torch.int32, device=dev)
    dense = torch.empty(NB * stride, dtype=torch.uint8, device=dev); doff = torch.empty(NB + 1, dtype=torch.int64, device=dev)
    ooff = torch.arange(NB + 1, dtype=torch.int64, device=dev) * BL
    out = torch.empty(NB * BL, dtype=torch.uint8, device=dev); res = torch.empty(NB, dtype=torch.int32, device=dev)
    eng.compress_batch_device(src, NB, BL, slots, stride, flen); eng.compact_device(slots, stride, flen, NB, dense, NB * stride, doff); eng.synchronize()
    C = int(doff[-1].item())
    e = [S.Event() for _ in range(2)]
    best = 1e9
    for it in range(6):
        eng.record(e[0]); eng.decompress_batch_device(dense, C, doff, NB, out, ooff, res); eng.record(e[1]); eng.synchronize()
        best = min(best, eng.elapsed_ms(e[0], e[1]))
    ok = bool((res == BL).all().item()) and torch.equal(out, src)
    print("%s block %7d x %6d: %.3f ms = %.0f GB/s ratio %.3f ok=%s" % (kind, BL, NB, best, NB * BL / best / 1e6, NB * BL / C, ok), flush=True)
    del src, slots, dense, out
