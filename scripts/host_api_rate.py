"""Development aid: PCIe-inclusive throughput of the host-buffer batched C API (what the Haskell shim calls),
driven through ctypes with preallocated buffers (no Python-side copies inside the timed region); pageable
caller memory (staged through pinned slots) and page-locked caller memory (handed to the DMA engines).
    python scripts/host_api_rate.py [out.jsonl]"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import numpy as np, torch
import streamly_lz4_amd as S
L = S.lib
eng = S.Engine(0)
BL, NB = 65536, 8192                     # 512 MiB per call
u8p, i32p, u64p = C.POINTER(C.c_uint8), C.POINTER(C.c_int32), C.POINTER(C.c_uint64)
dev = torch.device("cuda:0")
recs = []
for kind in ("lzsynth", "random"):
    src = torch.empty(NB * BL, dtype=torch.uint8, device=dev); eng.generate(kind, src, BL, NB); eng.synchronize()
    cap = NB * (S.compress_bound(BL) + 8)
    for mem in ("pageable", "pinned"):
        mk = (lambda n: torch.empty(n, dtype=torch.uint8).pin_memory()) if mem == "pinned" else (lambda n: torch.empty(n, dtype=torch.uint8))
        host_t, framed_t, out_t = mk(NB * BL), mk(cap), mk(NB * BL)
        host_t.copy_(src.cpu())
        host, framed, out = host_t.numpy(), framed_t.numpy(), out_t.numpy()
        ptrs = (u8p * NB)(*[C.cast(host.ctypes.data + i * BL, u8p) for i in range(NB)])
        lens = np.full(NB, BL, dtype=np.int32)
        flen = np.zeros(NB, dtype=np.int32); st = np.zeros(NB, dtype=np.int32); blen = np.zeros(NB, dtype=np.int32)
        olen = C.c_size_t(); got = C.c_int(); dlen = C.c_size_t()
        tc = td = 1e9
        for it in range(4):
            t0 = time.perf_counter()
            rc = L.mi355lz4_compress_batch(eng.ctx, ptrs, lens.ctypes.data_as(i32p), NB, 1, 8, framed.ctypes.data_as(u8p), cap, C.byref(olen),
                                           flen.ctypes.data_as(i32p), st.ctypes.data_as(i32p))
            t1 = time.perf_counter()
            assert rc == 0, S.lib.mi355lz4_last_error()
            rc = L.mi355lz4_decompress_batch(eng.ctx, framed.ctypes.data_as(u8p), olen.value, 8, 0, 0, None, 0, out.ctypes.data_as(u8p), out.size,
                                             C.byref(dlen), blen.ctypes.data_as(i32p), NB, C.byref(got))
            t2 = time.perf_counter()
            assert rc == 0 and dlen.value == NB * BL, S.lib.mi355lz4_last_error()
            tc, td = min(tc, t1 - t0), min(td, t2 - t1)
        assert np.array_equal(out, host)
        U = NB * BL
        rec = {"api": "host-buffer C API", "kind": kind, "caller_memory": mem, "MiB_per_call": U >> 20,
               "compress_GBps": round(U / tc / 1e9, 2), "decompress_GBps": round(U / td / 1e9, 2), "ratio": round(U / olen.value, 3)}
        recs.append(rec)
        print(rec, flush=True)
if len(sys.argv) > 1:
    with open(sys.argv[1], "a") as f:
        for r in recs:
            f.write(json.dumps(r) + "\n")
