"""Development aid: where the time of a SMALL call goes (the reference's own benchmark normalises its files to
10 MiB, benchmark/Main.hs:80-84): kernel alone, raw host-buffer C call (pageable / page-locked), Python Engine
wrapper, stream combinator.    python scripts/small_input_latency.py [MiB]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import numpy as np, torch
import streamly_lz4_amd as S
MiB = int(sys.argv[1]) if len(sys.argv) > 1 else 10
BL = 65536; NB = MiB * (1 << 20) // BL
dev = torch.device("cuda:0"); eng = S.Engine(0)
src = torch.empty(NB * BL, dtype=torch.uint8, device=dev); eng.generate("text", src, BL, NB); eng.synchronize()
raw = src.cpu().numpy().tobytes()
blocks = [raw[i * BL:(i + 1) * BL] for i in range(NB)]
fr, flen = eng.compress_batch(blocks)
u8p, i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int32)

def best(fn, n=20):
    t = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); fn(); t = min(t, time.perf_counter() - t0)
    return t * 1e3

# kernel alone (device-resident)
stride = S.slot_stride(BL, 8)
buf = torch.from_numpy(np.frombuffer(fr, dtype=np.uint8).copy()).to(dev)
offs = np.zeros(NB + 1, dtype=np.int64); np.cumsum(flen, out=offs[1:])
off = torch.from_numpy(offs).to(dev); ooff = torch.arange(NB + 1, dtype=torch.int64, device=dev) * BL
out = torch.empty(NB * BL, dtype=torch.uint8, device=dev); res = torch.empty(NB, dtype=torch.int32, device=dev)
e0, e1 = S.Event(), S.Event()
tk = 1e9
for _ in range(10):
    eng.record(e0); eng.decompress_batch_device(buf, len(fr), off, NB, out, ooff, res); eng.record(e1); eng.synchronize()
    tk = min(tk, eng.elapsed_ms(e0, e1))
print("input %d MiB (%d blocks): decode kernel alone %.3f ms" % (MiB, NB, tk))
for mem in ("pageable", "pinned"):
    mk = (lambda a: torch.from_numpy(a.copy()).pin_memory().numpy()) if mem == "pinned" else (lambda a: a.copy())
    framed = mk(np.frombuffer(fr, dtype=np.uint8)); o = mk(np.zeros(NB * BL, dtype=np.uint8))
    blen = np.zeros(NB, dtype=np.int32); dlen, got = C.c_size_t(), C.c_int()
    def call():
        rc = S.lib.mi355lz4_decompress_batch(eng.ctx, framed.ctypes.data_as(u8p), framed.size, 8, 0, 0, None, 0, o.ctypes.data_as(u8p),
                                             o.size, C.byref(dlen), blen.ctypes.data_as(i32p), NB, C.byref(got))
        assert rc == 0
    print("  raw C call, %s caller memory: %.3f ms" % (mem, best(call)))
    ptrs = (u8p * NB)(*[C.cast(o.ctypes.data + i * BL, u8p) for i in range(NB)])
    lens = np.full(NB, BL, dtype=np.int32); fl = np.zeros(NB, dtype=np.int32); st = np.zeros(NB, dtype=np.int32)
    cap = NB * (S.compress_bound(BL) + 8); fo = mk(np.zeros(cap, dtype=np.uint8)); olen = C.c_size_t()
    def ccall():
        rc = S.lib.mi355lz4_compress_batch(eng.ctx, ptrs, lens.ctypes.data_as(i32p), NB, 1, 8, fo.ctypes.data_as(u8p), cap, C.byref(olen),
                                           fl.ctypes.data_as(i32p), st.ctypes.data_as(i32p))
        assert rc == 0
    print("  raw C call, %s caller memory, compress: %.3f ms" % (mem, best(ccall)))
print("  Engine.decompress_batch (Python wrapper): %.3f ms" % best(lambda: eng.decompress_batch(fr)))
cfg = S.defaultBlockConfig
arrs = [fr[i:i + BL] for i in range(0, len(fr), BL)]
print("  decompressChunks combinator: %.3f ms" % best(lambda: S.decompressChunks(cfg, arrs, eng)))
print("  compressChunks combinator: %.3f ms" % best(lambda: S.compressChunks(cfg, 1, blocks, eng)))
# the C++ mirror alone (its C surface, arrays packed beforehand, results left in the C++ vectors)
data, lens, n = S._pack(arrs)
h = C.c_void_p()
def mirror_dec():
    rc = S.lib.slz4_decompress_chunks(eng._h, cfg.blockSize, 0, data.ctypes.data_as(u8p), lens.ctypes.data_as(C.POINTER(C.c_uint64)), n, C.byref(h))
    assert rc == 0
    S.lib.slz4_arrays_free(h)
print("  C++ mirror decompressChunks (no Python packing/unpacking): %.3f ms" % best(mirror_dec))
data2, lens2, n2 = S._pack(blocks)
def mirror_cmp():
    rc = S.lib.slz4_compress_chunks(eng._h, cfg.blockSize, 1, data2.ctypes.data_as(u8p), lens2.ctypes.data_as(C.POINTER(C.c_uint64)), n2, C.byref(h))
    assert rc == 0
    S.lib.slz4_arrays_free(h)
print("  C++ mirror compressChunks (no Python packing/unpacking): %.3f ms" % best(mirror_cmp))
