"""Quick end-to-end sanity + timing on a GPU box (development aid, not a test).

    python tests/gpu_sanity.py [n_blocks] [block_len]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))

import numpy as np
import torch

import streamly_lz4_amd as S
from oracle.oracle import Oracle

O = Oracle()
dev = torch.device("cuda:0")
eng = S.Engine(0)
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
BL = int(sys.argv[2]) if len(sys.argv) > 2 else 65536


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x)).to(dev)


# 1. generators
for kind in ("random", "lzsynth", "text"):
    buf = torch.empty(8 * 5000, dtype=torch.uint8, device=dev)
    eng.generate(kind, buf, 5000, 8, first_block=3)
    eng.synchronize()
    ref = O.gen(kind, 8, 5000, first_block=3)
    assert np.array_equal(buf.cpu().numpy(), ref), kind
print("generators ok")

# 2/3. round trip at small scale vs oracle
for kind in ("lzsynth", "text", "random"):
    for bl in (0, 1, 12, 13, 64, 1000, 65536, 100000):
        n = 6
        raw = O.gen(kind, n, max(bl, 1))[: n * bl] if bl else np.zeros(0, np.uint8)
        blocks = [raw[i * bl:(i + 1) * bl].tobytes() for i in range(n)]
        framed, flen = eng.compress_batch(blocks, accel=1)
        # oracle decodes our stream (independent and linked semantics)
        dec = O.frame_decompress(framed, n * bl, 8, 0, True)
        assert dec == raw.tobytes(), (kind, bl, "oracle decode of GPU stream")
        # GPU decodes oracle stream (independent blocks)
        ofr = O.frame_compress(raw.tobytes(), max(bl, 1), 1, 8, False) if bl else framed
        out, blen = eng.decompress_batch(ofr)
        assert out == raw.tobytes(), (kind, bl, "gpu decode of oracle stream", blen[:4])
        # GPU decodes reference-style linked stream
        lfr = O.frame_compress(raw.tobytes(), max(bl, 1), 1, 8, True) if bl else framed
        out, blen = eng.decompress_batch(lfr, linked=True)
        assert out == raw.tobytes(), (kind, bl, "gpu linked decode", blen[:4])
        osz = len(O.frame_compress(raw.tobytes(), max(bl, 1), 1, 8, False)) if bl else len(framed)
        print(kind, bl, "ok  gpu size %d vs oracle %d" % (len(framed), osz))

# 5. timing, device resident
for kind in ("lzsynth", "text", "random"):
    src = torch.empty(NB * BL, dtype=torch.uint8, device=dev)
    eng.generate(kind, src, BL, NB)
    stride = S.slot_stride(BL, 8)
    slots = torch.empty(NB * stride, dtype=torch.uint8, device=dev)
    flen = torch.empty(NB, dtype=torch.int32, device=dev)
    dense = torch.empty(NB * stride, dtype=torch.uint8, device=dev)
    doff = torch.empty(NB + 1, dtype=torch.int64, device=dev)
    ooff = torch.arange(NB + 1, dtype=torch.int64, device=dev) * BL
    out = torch.empty(NB * BL, dtype=torch.uint8, device=dev)
    res = torch.empty(NB, dtype=torch.int32, device=dev)
    e0, e1, e2, e3 = S.Event(), S.Event(), S.Event(), S.Event()
    for it in range(3):
        eng.record(e0)
        eng.compress_batch_device(src, NB, BL, slots, stride, flen, accel=1)
        eng.record(e1)
        eng.compact_device(slots, stride, flen, NB, dense, NB * stride, doff)
        eng.record(e2)
        eng.decompress_batch_device(dense, NB * stride, doff, NB, out, ooff, res)
        eng.record(e3)
        eng.synchronize()
    tc, tk, td = eng.elapsed_ms(e0, e1), eng.elapsed_ms(e1, e2), eng.elapsed_ms(e2, e3)
    total_c = int(doff[-1].item())
    ok = bool((res == BL).all().item()) and torch.equal(out, src)
    U = NB * BL
    print("%-8s NB=%d BL=%d ratio=%.3f  compress %.1f GB/s  compact %.1f GB/s(comp bytes)  decompress %.1f GB/s  (U+C)/t=%.1f GB/s  roundtrip_ok=%s"
          % (kind, NB, BL, U / total_c, U / tc / 1e6, total_c / tk / 1e6, U / td / 1e6, (U + total_c) / td / 1e6, ok))
    for v in (1, 2):
        eng.set_decoder(v)
        eng.record(e2)
        eng.decompress_batch_device(dense, NB * stride, doff, NB, out, ooff, res)
        eng.record(e3)
        eng.synchronize()
        print("   decoder variant %d: %.1f GB/s" % (v, U / eng.elapsed_ms(e2, e3) / 1e6))
    eng.set_decoder(0)
print("done")
