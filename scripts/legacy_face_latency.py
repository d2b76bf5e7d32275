"""Development aid: time per block through the 7-symbol legacy face (what an unmodified Streamly.Internal.LZ4 calls:
one block per call, src/Streamly/Internal/LZ4.hs:123-140).   python scripts/legacy_face_latency.py [kind] [blocks]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import numpy as np
import streamly_lz4_amd as S
from oracle.oracle import Oracle

kind = sys.argv[1] if len(sys.argv) > 1 else "text"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 160
BL = 65536
L = S.lib
u8p = C.POINTER(C.c_uint8)
L.LZ4_createStream.restype = C.c_void_p
L.LZ4_createStreamDecode.restype = C.c_void_p
L.LZ4_compress_fast_continue.argtypes = [C.c_void_p, u8p, u8p, C.c_int, C.c_int, C.c_int]
L.LZ4_decompress_safe_continue.argtypes = [C.c_void_p, u8p, u8p, C.c_int, C.c_int]
L.LZ4_freeStream.argtypes = [C.c_void_p]
L.LZ4_freeStreamDecode.argtypes = [C.c_void_p]
o = Oracle()
raw = o.gen(kind, nb, BL)
blocks = [raw[i * BL:(i + 1) * BL].copy() for i in range(nb)]
bound = S.compress_bound(BL)
comp = [np.zeros(bound, dtype=np.uint8) for _ in range(nb)]
back = [np.zeros(BL, dtype=np.uint8) for _ in range(nb)]
for rep in range(3):
    cctx, dctx = L.LZ4_createStream(), L.LZ4_createStreamDecode()
    t0 = time.perf_counter()
    sizes = [L.LZ4_compress_fast_continue(cctx, b.ctypes.data_as(u8p), c.ctypes.data_as(u8p), BL, bound, 1) for b, c in zip(blocks, comp)]
    t1 = time.perf_counter()
    got = [L.LZ4_decompress_safe_continue(dctx, c.ctypes.data_as(u8p), d.ctypes.data_as(u8p), n, BL) for c, d, n in zip(comp, back, sizes)]
    t2 = time.perf_counter()
    L.LZ4_freeStream(cctx); L.LZ4_freeStreamDecode(dctx)
assert all(n > 0 for n in sizes) and got == [BL] * nb and all(np.array_equal(a, b) for a, b in zip(blocks, back))
# the reference's own linked stream through the legacy decoder
fr = o.frame_compress(raw.tobytes(), BL, 1, 8, True)
dctx = L.LZ4_createStreamDecode()
pos, k, t3 = 0, 0, time.perf_counter()
while pos < len(fr):
    c = int.from_bytes(fr[pos:pos + 4], "little")
    blk = np.frombuffer(fr[pos + 8:pos + 8 + c], dtype=np.uint8)
    assert L.LZ4_decompress_safe_continue(dctx, blk.ctypes.data_as(u8p), back[k].ctypes.data_as(u8p), c, BL) == BL
    assert np.array_equal(back[k], blocks[k])
    pos += 8 + c; k += 1
t4 = time.perf_counter()
L.LZ4_freeStreamDecode(dctx)
print("%s, %d blocks of 64 KiB through the legacy face: compress %.3f ms/block (%.0f MB/s), decompress %.3f ms/block (%.0f MB/s); "
      "reference-written linked stream %.3f ms/block (incl. the check)" %
      (kind, nb, (t1 - t0) * 1e3 / nb, nb * BL / (t1 - t0) / 1e6, (t2 - t1) * 1e3 / nb, nb * BL / (t2 - t1) / 1e6, (t4 - t3) * 1e3 / nb))
