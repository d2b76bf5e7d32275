"""Development aid (GPU box): the two decoder forms and the default choice (variant 0) on streams that hardly compress -- long
literal runs end the workgroup form's segments.   python scripts/cu_decode_lowratio.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import numpy as np, torch, streamly_lz4_amd as S
from oracle.oracle import Oracle
O = Oracle(); eng = S.Engine(0)
nblk, BL = 160, 65536
for kind, accel in (("lzsynth", 65537), ("random", 1), ("lzsynth", 1), ("text", 64), ("text", 16), ("text", 8), ("text", 4), ("text", 2), ("lzsynth", 16), ("lzsynth", 64), ("zeros", 1), ("period7", 1), ("period300", 1), ("runs", 1), ("runs2", 1), ("runs3", 1)):
    if kind == "zeros":
        raw = bytes(nblk * BL)
    elif kind.startswith("period"):
        import random
        pat = random.Random(5).randbytes(int(kind[6:])); raw = (pat * (nblk * BL // len(pat) + 1))[: nblk * BL]
    elif kind in ("runs2", "runs3"):   # 64 / 200 random bytes, then a run of 530 to 1500 / 100 to 700 equal bytes (matches with three length bytes / two)
        import random
        rr = random.Random(7); parts = []
        while sum(map(len, parts)) < nblk * BL:
            parts.append(rr.randbytes(64 if kind == "runs2" else 200)); parts.append(bytes([rr.randrange(256)]) * (rr.randrange(530, 1500) if kind == "runs2" else rr.randrange(100, 700)))
        raw = b"".join(parts)[: nblk * BL]
    elif kind == "runs":            # runs of 40 to 4000 equal bytes between 8 random ones
        import random
        rr = random.Random(6); parts = []
        while sum(map(len, parts)) < nblk * BL:
            parts.append(rr.randbytes(8)); parts.append(bytes([rr.randrange(256)]) * rr.randrange(40, 4000))
        raw = b"".join(parts)[: nblk * BL]
    else:
        raw = O.gen(kind, nblk, BL, first_block=300).tobytes()
    blocks = [raw[i:i + BL] for i in range(0, len(raw), BL)]
    out_ = []
    for b in blocks:
        c = O.compress_block(b, accel)
        out_.append(len(c).to_bytes(4, "little") + len(b).to_bytes(4, "little") + c)
    fr = b"".join(out_)
    dev = torch.frombuffer(bytearray(fr), dtype=torch.uint8).cuda()
    offs, pos = [], 0
    for _ in range(nblk):
        offs.append(pos); pos += 8 + int.from_bytes(fr[pos:pos + 4], "little")
    boff = torch.tensor(offs + [pos], dtype=torch.int64, device="cuda")
    ooff = torch.arange(0, (nblk + 1) * BL, BL, dtype=torch.int64, device="cuda")
    out = torch.empty(nblk * BL, dtype=torch.uint8, device="cuda"); res = torch.zeros(nblk, dtype=torch.int32, device="cuda")
    line = "%-8s accel %-6d ratio %.3f" % (kind, accel, nblk * BL / len(fr))
    for v in (2, 4, 0):
        eng.set_decoder(v); best = 1e9
        for rep in range(10):
            e0, e1 = S.Event(), S.Event(); eng.record(e0)
            eng.decompress_batch_device(dev, len(fr), boff, nblk, out, ooff, res); eng.record(e1); torch.cuda.synchronize()
            best = min(best, eng.elapsed_ms(e0, e1))
        assert out.cpu().numpy().tobytes() == raw
        line += "  v%d %.4f ms" % (v, best)
    print(line, flush=True)
