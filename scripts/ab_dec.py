"""Development aid: time the decoder variants (and the encoder) of the built library on device-resident data.
    python scripts/ab_dec.py [kinds] [nblocks] [variants]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import torch, streamly_lz4_amd as S
kinds = (sys.argv[1] if len(sys.argv) > 1 else "lzsynth,text").split(",")
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
variants = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "2,3").split(",")]
dev = torch.device("cuda:0"); eng = S.Engine(0); BL = int(os.environ.get("AB_BL", "65536"))
for kind in kinds:
    src = torch.empty(NB * BL, dtype=torch.uint8, device=dev); eng.generate(kind, src, BL, NB)
    stride = S.slot_stride(BL, 8)
    slots = torch.empty(NB * stride, dtype=torch.uint8, device=dev); flen = torch.empty(NB, dtype=torch.int32, device=dev)
    dense = torch.empty(NB * stride, dtype=torch.uint8, device=dev); doff = torch.empty(NB + 1, dtype=torch.int64, device=dev)
    ooff = torch.arange(NB + 1, dtype=torch.int64, device=dev) * BL
    out = torch.empty(NB * BL, dtype=torch.uint8, device=dev); res = torch.empty(NB, dtype=torch.int32, device=dev)
    e = [S.Event() for _ in range(3)]
    tc = 1e9
    for it in range(3):
        eng.record(e[0]); eng.compress_batch_device(src, NB, BL, slots, stride, flen)
        eng.record(e[1]); eng.compact_device(slots, stride, flen, NB, dense, NB * stride, doff); eng.synchronize()
        tc = min(tc, eng.elapsed_ms(e[0], e[1]))
    C = int(doff[-1].item()); U = NB * BL
    line = "%s: enc %.0f GB/s ratio %.3f |" % (kind, U / tc / 1e6, U / C)
    for dv in variants:
        eng.set_decoder(dv); td = 1e9
        out.zero_()
        for it in range(6):
            eng.record(e[1]); eng.decompress_batch_device(dense, NB * stride, doff, NB, out, ooff, res); eng.record(e[2]); eng.synchronize()
            td = min(td, eng.elapsed_ms(e[1], e[2]))
        ok = bool((res == BL).all().item()) and torch.equal(out, src)
        line += " dec%d %.3f ms %.0f GB/s (U+C)/t/8TB %.4f ok=%s |" % (dv, td, U / td / 1e6, (U + C) / td / 1e6 / 8000, ok)
    print(line, flush=True)
