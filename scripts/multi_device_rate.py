"""Development aid: the multi-device handle of the C ABI (csrc/multi_device.cpp) against the single engine, host buffer to host
buffer, with 1, 2 and 3 engines ON ONE GPU (the box has one): a rehearsal -- the engines share one PCIe link here, so no
scaling is to be expected; what it shows is that the ranges run side by side and what the handle costs.
    python scripts/multi_device_rate.py [MiB]"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import numpy as np, torch, streamly_lz4_amd as S
MiB = int(sys.argv[1]) if len(sys.argv) > 1 else 256
BL = 65536; NB = MiB * 16
L = S.lib
u8p, i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
eng = S.Engine(0)
src = torch.empty(NB * BL, dtype=torch.uint8, device="cuda:0"); eng.generate("lzsynth", src, BL, NB); eng.synchronize()
host = src.cpu().pin_memory()
cap = NB * (S.compress_bound(BL) + 8)
framed = torch.empty(cap, dtype=torch.uint8).pin_memory(); out = torch.empty(NB * BL, dtype=torch.uint8).pin_memory()
ptrs = (u8p * NB)(*[C.cast(host.data_ptr() + i * BL, u8p) for i in range(NB)])
lens = np.full(NB, BL, dtype=np.int32); fl = np.zeros(NB, dtype=np.int32); st = np.zeros(NB, dtype=np.int32); bl = np.zeros(NB, dtype=np.int32)
olen, dlen, got = C.c_size_t(), C.c_size_t(), C.c_int()
def run(comp, decomp):
    tc = td = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); rc = comp(); t1 = time.perf_counter(); rc2 = decomp(); t2 = time.perf_counter()
        assert rc == 0 and rc2 == 0 and dlen.value == NB * BL, (rc, rc2)
        tc, td = min(tc, t1 - t0), min(td, t2 - t1)
    assert torch.equal(out, host)
    return round(NB * BL / tc / 1e9, 2), round(NB * BL / td / 1e9, 2)
c, d = run(lambda: L.mi355lz4_compress_batch(eng.ctx, ptrs, lens.ctypes.data_as(i32p), NB, 1, 8, C.cast(framed.data_ptr(), u8p), C.c_size_t(cap), C.byref(olen), fl.ctypes.data_as(i32p), st.ctypes.data_as(i32p)),
           lambda: L.mi355lz4_decompress_batch(eng.ctx, C.cast(framed.data_ptr(), u8p), C.c_size_t(olen.value), 8, 0, 0, None, 0, C.cast(out.data_ptr(), u8p), C.c_size_t(NB * BL), C.byref(dlen), bl.ctypes.data_as(i32p), NB, C.byref(got)))
print(json.dumps({"handle": "single engine", "MiB": MiB, "memory": "pinned", "compress_GBps": c, "decompress_GBps": d}), flush=True)
for n in (1, 2, 3):
    m = S.MultiEngine([0] * n)
    c, d = run(lambda: L.mi355lz4_multi_compress_batch(m._h, ptrs, lens.ctypes.data_as(i32p), NB, 1, 8, C.cast(framed.data_ptr(), u8p), C.c_size_t(cap), C.byref(olen), fl.ctypes.data_as(i32p), st.ctypes.data_as(i32p)),
               lambda: L.mi355lz4_multi_decompress_batch(m._h, C.cast(framed.data_ptr(), u8p), C.c_size_t(olen.value), 8, 0, C.cast(out.data_ptr(), u8p), C.c_size_t(NB * BL), C.byref(dlen), bl.ctypes.data_as(i32p), NB, C.byref(got)))
    print(json.dumps({"handle": "multi, %d engine(s) on device 0" % n, "MiB": MiB, "memory": "pinned", "compress_GBps": c, "decompress_GBps": d,
                      "note": "one GPU, one link: a rehearsal, not a scaling figure"}), flush=True)
    m.close()
