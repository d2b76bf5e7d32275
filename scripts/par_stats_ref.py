"""Development aid (review item 6): the lane-parallel decoder on the SAME data written by the engine's compressor and by the
reference's (independent blocks both): rate and the STATS kernel's counters side by side, for every library variant under
lib/variants/ (MI355LZ4_LIB).  The streams are built once and cached in /tmp.
    python scripts/par_stats_ref.py [kind] [n_blocks]"""
import ctypes as C
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
kind = sys.argv[1] if len(sys.argv) > 1 else "lzsynth"
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
BL = 65536
cache = "/tmp/par_stats_ref_%s_%d" % (kind, NB)

if os.environ.get("PSR_CHILD") != "1":
    if not os.path.exists(cache + ".ref"):
        import numpy as np
        import streamly_lz4_amd as S
        from oracle.oracle import Oracle, Reference, have_reference
        import torch
        eng = S.Engine(0)
        src = torch.empty(NB * BL, dtype=torch.uint8, device="cuda:0")
        eng.generate(kind, src, BL, NB); eng.synchronize()
        raw = src.cpu().numpy().tobytes()
        blocks = [raw[i * BL:(i + 1) * BL] for i in range(NB)]
        codec = Reference() if have_reference() else Oracle()
        ref = b"".join((lambda c, b: len(c).to_bytes(4, "little") + len(b).to_bytes(4, "little") + c)(codec.compress_block(b, 1), b) for b in blocks)
        own = eng.compress_batch(blocks)[0]
        open(cache + ".raw", "wb").write(raw); open(cache + ".ref", "wb").write(ref); open(cache + ".own", "wb").write(own)
        eng.close()
    main = os.path.join(ROOT, "streamly-lz4_amd", "lib", "libmi355lz4.so")
    libs = [main] + sorted(glob.glob(os.path.join(ROOT, "streamly-lz4_amd", "lib", "variants", "*.so"))) + [main]
    for lib in libs:
        r = subprocess.run([sys.executable, __file__, kind, str(NB)], env=dict(os.environ, MI355LZ4_LIB=lib, PSR_CHILD="1"),
                           capture_output=True, text=True)
        print("%-30s %s" % (os.path.basename(lib), r.stdout.strip() or r.stderr.strip()[-400:]), flush=True)
    sys.exit(0)

import numpy as np
import torch
import streamly_lz4_amd as S
NAMES = ["batches", "seqs", "rounds", "match_iters", "lit_iters", "handovers", "slides", "full", "far",
         "t_window", "t_spec", "t_chain", "t_decode", "t_lit", "t_need", "t_match", "t_flush", "t_seq"]
S.lib.mi355lz4_debug_stats.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
eng = S.Engine(0)
eng.set_decoder(2)
raw = torch.frombuffer(bytearray(open(cache + ".raw", "rb").read()), dtype=torch.uint8).cuda()
line = []
for who in ("own", "ref"):
    fr = open(cache + "." + who, "rb").read()
    offs, pos = np.zeros(NB + 1, dtype=np.int64), 0
    for i in range(NB):
        offs[i] = pos
        pos += 8 + int.from_bytes(fr[pos:pos + 4], "little")
    offs[NB] = pos
    dev = torch.frombuffer(bytearray(fr), dtype=torch.uint8).cuda()
    boff = torch.from_numpy(offs).cuda()
    ooff = torch.arange(NB + 1, dtype=torch.int64, device="cuda") * BL
    out = torch.empty(NB * BL, dtype=torch.uint8, device="cuda")
    res = torch.empty(NB, dtype=torch.int32, device="cuda")
    e0, e1 = S.Event(), S.Event()
    best = 1e9
    for _ in range(6):
        eng.record(e0); eng.decompress_batch_device(dev, len(fr), boff, NB, out, ooff, res); eng.record(e1); eng.synchronize()
        best = min(best, eng.elapsed_ms(e0, e1))
    ok = bool((res == BL).all().item()) and torch.equal(out, raw)
    buf = (C.c_uint64 * 32)()
    S.lib.mi355lz4_debug_stats(eng.ctx, 1, buf)
    eng.decompress_batch_device(dev, len(fr), boff, NB, out, ooff, res); eng.synchronize()
    S.lib.mi355lz4_debug_stats(eng.ctx, 0, buf)
    st = dict(zip(NAMES, list(buf)))
    b = max(st["batches"], 1)
    line.append("%s %.0f GB/s ratio %.3f ok=%s [seq/blk %.0f batches/blk %.1f seq/batch %.1f rounds %.2f miters %.1f handovers/blk %.2f far/batch %.1f full %.0f%%]"
                % (who, NB * BL / best / 1e6, NB * BL / len(fr), ok, st["seqs"] / NB, b / NB, st["seqs"] / b, st["rounds"] / b,
                   st["match_iters"] / b, st["handovers"] / NB, st["far"] / b, 100 * st["full"] / b))
print(" | ".join(line))
