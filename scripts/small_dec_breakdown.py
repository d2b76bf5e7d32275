"""Development aid: where a 10 MiB decompressChunks call (the reference's benchmark protocol) spends its time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import numpy as np, torch, ctypes as C, streamly_lz4_amd as S
eng = S.Engine(0)
N = 10 << 20; BL = 65536
t = torch.empty(N, dtype=torch.uint8, device="cuda:0"); eng.generate("text", t, BL, N // BL); eng.synchronize()
raw = t.cpu().numpy().tobytes()
cfg = S.defaultBlockConfig
for accel in (1, 65537):
    framed = b"".join(S.compressChunks(cfg, accel, [raw[i:i + BL] for i in range(0, N, BL)], eng))
    def best(f, n=7):
        b = 1e9
        for _ in range(n):
            t0 = time.perf_counter(); r = f(); b = min(b, time.perf_counter() - t0)
        return b * 1e3, r
    t_chunks, chunks = best(lambda: [framed[i:i + BL] for i in range(0, len(framed), BL)])
    t_pack, packed = best(lambda: S._pack(chunks))
    data, lens, n = packed
    def ccall():
        h = C.c_void_p()
        rc = S.lib.slz4_decompress_chunks(eng._h, cfg.blockSize, 0, data.ctypes.data_as(S._u8p), lens.ctypes.data_as(S._u64p), n, C.byref(h))
        assert rc == 0
        return h
    t_c, h = best(ccall)
    t_views, v = best(lambda: S._unpack_views(ccall()), 3)
    t_all, out = best(lambda: S.decompressChunks(cfg, [framed[i:i + BL] for i in range(0, len(framed), BL)], eng, views=True))
    t_bytes, out2 = best(lambda: S.decompressChunks(cfg, [framed[i:i + BL] for i in range(0, len(framed), BL)], eng))
    # the C ABI alone, host buffers
    src = np.frombuffer(framed, dtype=np.uint8); outb = np.empty(N + 16, dtype=np.uint8)
    blen = np.zeros(N // BL + 1, dtype=np.int32); ol = C.c_size_t(); got = C.c_int()
    def cabi():
        rc = S.lib.mi355lz4_decompress_batch(eng.ctx, src.ctypes.data_as(S._u8p), src.size, 8, 0, 1, None, 0, outb.ctypes.data_as(S._u8p), N + 16,
                                             C.byref(ol), blen.ctypes.data_as(S._i32p), N // BL, C.byref(got))
        assert rc == 0
    t_abi, _ = best(cabi)
    print({"accel": accel, "framed_MB": round(len(framed) / 1e6, 2), "chunks_of_ms": round(t_chunks, 3), "pack_ms": round(t_pack, 3),
           "c_shim_ms": round(t_c, 3), "c_shim+views_ms": round(t_views, 3), "c_abi_host_ms": round(t_abi, 3),
           "decompressChunks_views_ms": round(t_all, 3), "decompressChunks_bytes_ms": round(t_bytes, 3), "ok": b"".join(out) == raw})
