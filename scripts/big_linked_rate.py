"""Development aid (GPU box): linked streams of big blocks -- the workgroup form against guessed dictionaries (api.cpp path 6) and the pointer pass."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import numpy as np, torch, streamly_lz4_amd as S
from oracle.oracle import Oracle
O = Oracle(); eng = S.Engine(0); dev = torch.device("cuda:0")
cases = ((1 << 20, 256), (4 << 20, 64), (512 << 10, 512))
big = None
for a_ in sys.argv[1:]:                      # big_linked_rate.py [case=<block KiB>x<blocks>] [big=<KiB from which the path is taken>]
    if a_.startswith("case="):
        k_, n_ = a_[5:].split("x"); cases = ((int(k_) << 10, int(n_)),)
    if a_.startswith("big="):
        big = a_[4:]
for bl, nblk in cases:
    raw = O.gen("text", nblk * bl // 65536, 65536, first_block=21).tobytes()
    if "data=mixz" in sys.argv:              # 32 KiB of text, 96 KiB of zeros, ...: big blocks that compress 7 x (armed behind the first pass)
        t_ = np.frombuffer(raw, dtype=np.uint8).copy().reshape(-1, 131072); t_[:, 32768:] = 0; raw = t_.tobytes()
    if "writer=engine" in sys.argv:          # the engine's own linked compressor (more of a block comes from the block before it)
        e2 = S.Engine(0); e2.set_linked_compress(True)
        srct = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).to(dev)
        stride = S.slot_stride(bl, 8); slots = torch.empty(nblk * stride, dtype=torch.uint8, device=dev); flen = torch.empty(nblk, dtype=torch.int32, device=dev)
        dense = torch.empty(nblk * stride, dtype=torch.uint8, device=dev); doff = torch.empty(nblk + 1, dtype=torch.int64, device=dev)
        e2.compress_batch_device(srct, nblk, bl, slots, stride, flen); e2.compact_device(slots, stride, flen, nblk, dense, nblk * stride, doff); e2.synchronize()
        fr = dense[: int(doff[-1].item())].cpu().numpy().tobytes(); e2.close(); del slots, dense, srct
    else:
        fr = O.frame_compress(raw, bl, 1, 8, True)
    offs, pos = [], 0
    for _ in range(nblk):
        offs.append(pos); pos += 8 + int.from_bytes(fr[pos:pos + 4], "little")
    buf = torch.frombuffer(bytearray(fr), dtype=torch.uint8).to(dev)
    boff = torch.tensor(offs + [pos], dtype=torch.int64, device=dev)
    ooff = torch.arange(0, (nblk + 1) * bl, bl, dtype=torch.int64, device=dev)
    out = torch.zeros(nblk * bl, dtype=torch.uint8, device=dev); res = torch.zeros(nblk, dtype=torch.int32, device=dev)
    for env in (big, "0"):
        if env is None: os.environ.pop("MI355LZ4_LINKED_BIG", None)
        else: os.environ["MI355LZ4_LINKED_BIG"] = env
        best = 1e9
        for rep in range(4):
            e0, e1 = S.Event(), S.Event(); eng.record(e0)
            eng.decompress_batch_device(buf, len(fr), boff, nblk, out, ooff, res, linked=True); eng.record(e1); eng.synchronize()
            best = min(best, eng.elapsed_ms(e0, e1))
        st = (C.c_int * 5)(); S.lib.mi355lz4_debug_runin_state(eng.ctx, st, None)
        ok = out.cpu().numpy().tobytes() == raw
        print("%d blocks of %d KiB linked text: LINKED_BIG=%s  %.3f ms  %.1f GB/s  path %d passes %d ok %s" % (nblk, bl >> 10, env, best, nblk * bl / best / 1e6, st[4], st[3], ok), flush=True)
