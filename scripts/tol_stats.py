"""Development aid: deferred-list statistics of the tolerant pass on a reference-linked text stream."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import numpy as np, torch, streamly_lz4_amd as S
from oracle.oracle import Oracle
O = Oracle(); eng = S.Engine(0); dev = torch.device("cuda:0")
bl, nb = 65536, 256
for kind in ("text", "lzsynth"):
    data = O.gen(kind, nb, bl, first_block=7).tobytes() if kind == "text" else None
    if kind == "lzsynth":
        base = O.gen("lzsynth", 2, bl, first_block=3).tobytes(); rng = np.random.default_rng(5)
        data = b"".join(base[int(o):int(o) + bl] for o in rng.integers(0, bl, nb))
    fr = O.frame_compress(data, bl, 1, 8, True)
    offs, pos = [], 0
    for _ in range(nb):
        offs.append(pos); pos += 8 + int.from_bytes(fr[pos:pos + 4], "little")
    buf = torch.from_numpy(np.frombuffer(fr, dtype=np.uint8).copy()).to(dev)
    off = torch.tensor(offs + [pos], dtype=torch.int64, device=dev)
    ooff = torch.arange(nb + 1, dtype=torch.int64, device=dev) * bl
    out = torch.zeros(nb * bl, dtype=torch.uint8, device=dev); res = torch.zeros(nb, dtype=torch.int32, device=dev)
    eng.decompress_batch_device(buf, len(fr), off, nb, out, ooff, res, linked=True); eng.synchronize()
    st = (C.c_longlong * 8)()
    S.lib.mi355lz4_debug_tol_stats.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_longlong)]
    rc = S.lib.mi355lz4_debug_tol_stats(eng.ctx, nb, st)
    meta = (C.c_int32 * 4)()
    print(kind, "rc", rc, "blocks with list %d, entries %d (avg %.0f), longest %d, overflowed %d, without list %d; ok=%s"
          % (st[0], st[1], st[1] / max(st[0], 1), st[2], st[3], st[4], bool((res == bl).all().item())))
    if st[6]:
        print("   replay: %d super-batches, %.1f rounds each, %.0f cycles per super-batch, %.0f cycles per block" % (st[6], st[5] / st[6], 16.0 * st[7] / st[6], 16.0 * st[7] / max(st[0], 1)))
