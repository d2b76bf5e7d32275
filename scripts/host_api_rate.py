"""Development aid: PCIe-inclusive throughput of the host-buffer batched C API (what the Haskell shim calls),
driven through ctypes with preallocated numpy buffers (no Python-side copies inside the timed region)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import numpy as np, torch
import streamly_lz4_amd as S
L = S.lib
eng = S.Engine(0)
BL, NB = 65536, 8192                     # 512 MiB per call
u8p, i32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_uint64)
dev = torch.device("cuda:0")
for kind in ("lzsynth", "random"):
    src = torch.empty(NB * BL, dtype=torch.uint8, device=dev); eng.generate(kind, src, BL, NB); eng.synchronize()
    host = src.cpu().numpy()
    ptrs = (u8p * NB)(*[C.cast(host.ctypes.data + i * BL, u8p) for i in range(NB)])
    lens = np.full(NB, BL, dtype=np.int32)
    cap = NB * (S.compress_bound(BL) + 8)
    framed = np.empty(cap, dtype=np.uint8); flen = np.zeros(NB, dtype=np.int32); st = np.zeros(NB, dtype=np.int32)
    out = np.empty(NB * BL, dtype=np.uint8); blen = np.zeros(NB, dtype=np.int32)
    olen = C.c_size_t(); got = C.c_int(); dlen = C.c_size_t()
    for it in range(3):
        t0 = time.perf_counter()
        rc = L.mi355lz4_compress_batch(eng.ctx, ptrs, lens.ctypes.data_as(i32p), NB, 1, 8, framed.ctypes.data_as(u8p), cap, C.byref(olen),
                                       flen.ctypes.data_as(i32p), st.ctypes.data_as(i32p))
        t1 = time.perf_counter()
        assert rc == 0
        rc = L.mi355lz4_decompress_batch(eng.ctx, framed.ctypes.data_as(u8p), olen.value, 8, 0, 0, None, 0, out.ctypes.data_as(u8p), out.size,
                                         C.byref(dlen), blen.ctypes.data_as(i32p), NB, C.byref(got))
        t2 = time.perf_counter()
        assert rc == 0 and dlen.value == NB * BL
    assert np.array_equal(out, host)
    U = NB * BL
    print("%s: host-buffer C API, %d MiB per call, pageable buffers: compress %.2f GB/s, decompress %.2f GB/s (ratio %.2f)"
          % (kind, U >> 20, U / (t1 - t0) / 1e9, U / (t2 - t1) / 1e9, U / olen.value))
