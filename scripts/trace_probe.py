import ctypes as C, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import numpy as np, torch
import streamly_lz4_amd as S
L = S.lib; eng = S.Engine(0); BL, NB = 65536, 8192
u8p, i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
dev = torch.device("cuda:0")
src = torch.empty(NB * BL, dtype=torch.uint8, device=dev); eng.generate("lzsynth", src, BL, NB); eng.synchronize()
cap = NB * (S.compress_bound(BL) + 8)
host_t = torch.empty(NB * BL, dtype=torch.uint8).pin_memory(); framed_t = torch.empty(cap, dtype=torch.uint8).pin_memory()
host_t.copy_(src.cpu())
ptrs = (u8p * NB)(*[C.cast(host_t.data_ptr() + i * BL, u8p) for i in range(NB)])
lens = np.full(NB, BL, dtype=np.int32); flen = np.zeros(NB, dtype=np.int32); st = np.zeros(NB, dtype=np.int32); olen = C.c_size_t()
for it in range(3):
    sys.stderr.write("---- call %d\n" % it)
    t0 = time.perf_counter()
    rc = L.mi355lz4_compress_batch(eng.ctx, ptrs, lens.ctypes.data_as(i32p), NB, 1, 8, C.cast(framed_t.data_ptr(), u8p), cap, C.byref(olen), flen.ctypes.data_as(i32p), st.ctypes.data_as(i32p))
    sys.stderr.write("call took %.2f ms\n" % ((time.perf_counter() - t0) * 1e3))
