"""Development aid: phase breakdown of the lane-parallel decoder (STATS kernel).
    python scripts/par_stats.py [kind] [n_blocks]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import torch
import streamly_lz4_amd as S

NAMES = ["batches", "seqs", "rounds", "match_iters", "lit_iters", "handovers", "slides", "full", "far",
         "t_window", "t_spec", "t_chain", "t_decode", "t_lit", "t_need", "t_match", "t_flush", "t_seq"]
kinds = [sys.argv[1]] if len(sys.argv) > 1 else ["lzsynth", "text"]
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
BL = 65536
dev = torch.device("cuda:0")
eng = S.Engine(0)
S.lib.mi355lz4_debug_stats.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
for kind in kinds:
    src = torch.empty(NB * BL, dtype=torch.uint8, device=dev)
    eng.generate(kind, src, BL, NB)
    stride = S.slot_stride(BL, 8)
    slots = torch.empty(NB * stride, dtype=torch.uint8, device=dev)
    flen = torch.empty(NB, dtype=torch.int32, device=dev)
    dense = torch.empty(NB * stride, dtype=torch.uint8, device=dev)
    doff = torch.empty(NB + 1, dtype=torch.int64, device=dev)
    ooff = torch.arange(NB + 1, dtype=torch.int64, device=dev) * BL
    out = torch.empty(NB * BL, dtype=torch.uint8, device=dev)
    res = torch.empty(NB, dtype=torch.int32, device=dev)
    eng.compress_batch_device(src, NB, BL, slots, stride, flen)
    eng.compact_device(slots, stride, flen, NB, dense, NB * stride, doff)
    eng.synchronize()
    buf = (C.c_uint64 * 32)()
    S.lib.mi355lz4_debug_stats(eng.ctx, 1, buf)
    eng.decompress_batch_device(dense, NB * stride, doff, NB, out, ooff, res)
    eng.synchronize()
    S.lib.mi355lz4_debug_stats(eng.ctx, 0, buf)
    st = dict(zip(NAMES, list(buf)))
    b = max(st["batches"], 1)
    tt = sum(v for k, v in st.items() if k.startswith("t_"))
    print("== %s: %d blocks; per block: %.1f batches, %.1f seq/batch, %.2f rounds/batch, %.1f match iters/batch, %.1f lit iters/batch, "
          "handovers/blk %.2f, slides/blk %.1f, full batches %.0f%%, far lanes/batch %.1f"
          % (kind, NB, b / NB, st["seqs"] / b, st["rounds"] / b, st["match_iters"] / b, st["lit_iters"] / b,
             st["handovers"] / NB, st["slides"] / NB, 100 * st["full"] / b, st["far"] / b))
    print("   cycles/batch: " + "  ".join("%s %.0f (%.0f%%)" % (k[2:], st[k] / b, 100 * st[k] / tt) for k in NAMES if k.startswith("t_")))
    print("   total cycles/block %.0f  -> %.2f bytes/cycle/wave" % (tt / NB, BL * NB / tt))
