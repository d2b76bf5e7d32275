"""Development aid: the dictionary share the run-in decision samples (api.cpp, k_dict_share) for the streams it has to tell apart.
    python scripts/runin_share.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import numpy as np, torch, streamly_lz4_amd as S
from oracle.oracle import Oracle
O = Oracle(); BL = 65536; NB = 9472
dev = torch.device("cuda:0")
def run(name, framed, raw):
    eng = S.Engine(0)
    offs, pos = np.zeros(NB + 1, dtype=np.int64), 0
    for i in range(NB):
        offs[i] = pos; pos += 8 + int.from_bytes(framed[pos:pos + 4], "little")
    offs[NB] = pos
    buf = torch.from_numpy(np.frombuffer(framed, dtype=np.uint8).copy()).to(dev); off = torch.from_numpy(offs).to(dev)
    ooff = torch.arange(NB + 1, dtype=torch.int64, device=dev) * BL
    out = torch.zeros(NB * BL, dtype=torch.uint8, device=dev); res = torch.zeros(NB, dtype=torch.int32, device=dev)
    eng.decompress_batch_device(buf, len(framed), off, NB, out, ooff, res, linked=True); eng.synchronize()
    st = (C.c_int * 5)(); S.lib.mi355lz4_debug_runin_state(eng.ctx, st, None)
    ok = out.cpu().numpy().tobytes() == raw
    print("%-44s sampled share %.4f  state %s  path %d  ok %s" % (name, st[3] / 1e6, list(st)[:3], st[4], ok), flush=True)
    eng.close()
def engine_linked(raw):
    e = S.Engine(0); e.set_linked_compress(True)
    src = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).to(dev)
    stride = S.slot_stride(BL, 8); slots = torch.empty(NB * stride, dtype=torch.uint8, device=dev); flen = torch.empty(NB, dtype=torch.int32, device=dev)
    dense = torch.empty(NB * stride, dtype=torch.uint8, device=dev); doff = torch.empty(NB + 1, dtype=torch.int64, device=dev)
    e.compress_batch_device(src, NB, BL, slots, stride, flen); e.compact_device(slots, stride, flen, NB, dense, NB * stride, doff); e.synchronize()
    fr = dense[: int(doff[-1].item())].cpu().numpy().tobytes(); e.close()
    return fr
for seed in (0, 7777):
    raw = O.gen("text", NB, BL, first_block=seed).tobytes()
    run("reference, linked, text (seed %d)" % seed, O.frame_compress(raw, BL, 1, 8, True), raw)
    run("engine, linked, text (seed %d)" % seed, engine_linked(raw), raw)
import glob
py = b"".join(open(f, "rb").read() for f in sorted(glob.glob("/usr/lib/python3.10/*.py")))
raw = (py * (NB * BL // len(py) + 1))[: NB * BL]
run("reference, linked, Python sources cycled", O.frame_compress(raw, BL, 1, 8, True), raw)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_linked_rate_gpu import _copying_stream
raw, fr = _copying_stream(NB, BL)
run("hand-written: every block copies the one before", fr, raw)
