#!/usr/bin/env python3
"""Development aid (GPU box): the workgroup-per-block decoder (decoder variant 4, csrc/decode_cu.hpp) against the
lane-parallel one (variant 2) -- same bytes, same per-block results (negative codes included) -- and the time of a small
device-resident call with either.

usage: cu_decode_check.py [check] [time] [blocks=160]
"""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import streamly_lz4_amd as S  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

O = Oracle()
eng = S.Engine(0)
args = sys.argv[1:]
nblk = 160
BL = 65536
for a in args:
    if a.startswith("blocks="):
        nblk = int(a.split("=")[1])
    if a.startswith("bl="):
        BL = int(a.split("=")[1])


def frame_ref(blocks):
    """independent blocks written by the oracle (= the reference's bytes), 8-byte headers"""
    out = []
    for b in blocks:
        c = O.compress_block(b, 1)
        out.append(len(c).to_bytes(4, "little") + len(b).to_bytes(4, "little") + c)
    return b"".join(out)


def both(framed, expect=None, what=""):
    res = {}
    for v in (2, 4):
        eng.set_decoder(v)
        out, blen = eng.decompress_batch(framed, raise_on_block_error=False)
        res[v] = (out, blen)
    eng.set_decoder(0)
    if res[2][1] != res[4][1]:
        bad = [(i, a, b) for i, (a, b) in enumerate(zip(res[2][1], res[4][1])) if a != b]
        raise SystemExit(f"{what}: per-block results differ (block, variant 2, variant 4): {bad[:8]}")
    if res[2][0] != res[4][0]:
        a, b = np.frombuffer(res[2][0], np.uint8), np.frombuffer(res[4][0], np.uint8)
        d = np.nonzero(a != b)[0]
        raise SystemExit(f"{what}: bytes differ at {d[:8]} ({d.size} of {a.size})")
    if expect is not None and res[4][0] != expect:
        a, b = np.frombuffer(expect, np.uint8), np.frombuffer(res[4][0], np.uint8)
        n = min(a.size, b.size)
        d = np.nonzero(a[:n] != b[:n])[0]
        raise SystemExit(f"{what}: wrong bytes: sizes {a.size} / {b.size}, first differences {d[:8]}")
    return res[4]


def check():
    rng = random.Random(7)
    for kind in ("lzsynth", "text", "random"):
        raw = O.gen(kind, 24, 65536, first_block=100).tobytes()
        blocks = [raw[i:i + 65536] for i in range(0, len(raw), 65536)]
        both(frame_ref(blocks), raw, f"{kind} reference-written")
        fr, _ = eng.compress_batch(blocks)
        both(fr, raw, f"{kind} engine-written")
        print("ok", kind, flush=True)
    # odd shapes: short, ragged, long runs, long literal stretches inside compressible data, small offsets
    from test_fuzz_encode_gpu import _make
    blocks = [_make(rng, O, t) for t in range(200)]
    blocks += [bytes(65536), bytes(40000), b"ab" * 30000, bytes(range(256)) * 200, O.gen("text", 1, 70000).tobytes(),
               O.gen("text", 1, 30000).tobytes() + O.gen("random", 1, 5000).tobytes() + O.gen("text", 1, 30000, first_block=9).tobytes(),
               O.gen("lzsynth", 1, 262144).tobytes(), O.gen("text", 1, 1000).tobytes(), b"", b"x", O.gen("text", 1, 65535).tobytes(),
               (O.gen("text", 1, 3000).tobytes() + bytes(1500)) * 14]
    raw = b"".join(blocks)
    both(frame_ref(blocks), raw, "odd shapes reference-written")
    fr, _ = eng.compress_batch(blocks)
    both(fr, raw, "odd shapes engine-written")
    print("ok odd shapes", flush=True)
    # corrupted blocks: the two variants must report the same codes (variant 4 leaves every failing block to the exact path)
    base = [O.gen("text", 1, 65536, first_block=5).tobytes(), O.gen("lzsynth", 1, 65536, first_block=6).tobytes()]
    nbad = 0
    for trial in range(300):
        b = base[trial & 1]
        c = bytearray(O.compress_block(b, 1))
        for _ in range(rng.choice((1, 1, 2, 5))):
            pos = rng.randrange(len(c)) if trial % 3 else rng.randrange(max(1, len(c) - 200), len(c))
            c[pos] = rng.randrange(256)
        cl = len(c) if trial % 7 else len(c) - rng.randrange(1, 40)
        ul = len(b) if trial % 5 else len(b) - rng.randrange(0, 300)
        fr = cl.to_bytes(4, "little") + ul.to_bytes(4, "little") + bytes(c[:cl])
        out, blen = both(fr, None, f"corrupted {trial}")
        code, _ = O.decompress_block(bytes(c[:cl]), ul)
        if blen[0] != code:
            raise SystemExit(f"corrupted {trial}: engine {blen[0]} oracle {code}")
        nbad += code < 0
    print("ok corrupted (", nbad, "of 300 rejected )", flush=True)


def timing():
    for kind in ("lzsynth", "text"):
        if BL > 65536:      # (the oracle's generators make 64 KiB blocks of their own seed: a big block is a run of them)
            raw = O.gen(kind, nblk * BL // 65536, 65536, first_block=300).tobytes()
        else:
            raw = O.gen(kind, nblk, BL, first_block=300).tobytes()
        blocks = [raw[i:i + BL] for i in range(0, len(raw), BL)]
        for writer in ("reference", "engine"):
            fr = frame_ref(blocks) if writer == "reference" else eng.compress_batch(blocks)[0]
            dev = torch.frombuffer(bytearray(fr), dtype=torch.uint8).cuda()
            offs, pos = [], 0
            for _ in range(nblk):
                offs.append(pos)
                pos += 8 + int.from_bytes(fr[pos:pos + 4], "little")
            boff = torch.tensor(offs + [pos], dtype=torch.int64, device="cuda")
            ooff = torch.arange(0, (nblk + 1) * BL, BL, dtype=torch.int64, device="cuda")
            out = torch.empty(nblk * BL, dtype=torch.uint8, device="cuda")
            res = torch.zeros(nblk, dtype=torch.int32, device="cuda")
            line = {"kind": kind, "writer": writer, "blocks": nblk}
            for v in (2, 4):
                eng.set_decoder(v)
                best = 1e9
                for rep in range(12 if nblk * BL <= (64 << 20) else 4):
                    out.zero_()
                    e0, e1 = S.Event(), S.Event()
                    eng.record(e0)
                    eng.decompress_batch_device(dev, len(fr), boff, nblk, out, ooff, res)
                    eng.record(e1)
                    torch.cuda.synchronize()
                    best = min(best, eng.elapsed_ms(e0, e1))
                assert out.cpu().numpy().tobytes() == raw and bool((res == BL).all()), (kind, writer, v)
                line[f"variant{v}_ms"] = round(best, 4)
                line[f"variant{v}_GBps"] = round(nblk * BL / best / 1e6, 1)
            # why / where the time goes (variant 4): 16 words per block, decode_cu.hpp `dbg`
            import ctypes as C
            dbg = torch.zeros(nblk * 16, dtype=torch.int32, device="cuda")
            S.lib.mi355lz4_debug_cu(eng.ctx, C.c_void_p(dbg.data_ptr()))
            eng.set_decoder(4)
            eng.decompress_batch_device(dev, len(fr), boff, nblk, out, ooff, res)
            torch.cuda.synchronize()
            S.lib.mi355lz4_debug_cu(eng.ctx, None)
            d = dbg.cpu().numpy().astype(np.int64).reshape(nblk, 16) & 0xffffffff
            line["why"] = np.bincount(d[:, 0], minlength=7).tolist()
            if "verbose" in args:
                from cu_decode_sim import parse as true_parse
                shown = 0
                for b in range(nblk):
                    if d[b, 0] != 0 and shown < 4:
                        comp = fr[offs[b] + 8: (offs + [pos])[b + 1]]
                        seqs = true_parse(comp)
                        inLim = len(comp) - 32
                        op = 0; exp = None
                        for i, (tp, ls, lit, off, ml, nx) in enumerate(seqs):
                            if not (ml and nx <= inLim and op + lit >= off and op + lit + ml + 64 < BL and i < 8192):
                                exp = (i, tp, op); break
                            op += lit + ml
                        print("  block", b, "why", d[b, 0], "nPar/tailIp/tailOp", d[b, 1:4].tolist(), "expected", exp, "C", len(comp), flush=True)
                        shown += 1
            line["nPar_mean"] = float((d[:, 1] & 0xffff).mean())
            line["rounds_median"] = int(np.median(d[:, 1] >> 16))
            ok = d[:, 0] == 0
            if ok.any():
                t = d[ok][:, 4:16]
                names = ["stage", "T", "cands", "rank", "entries", "records", "literals", "rankrec", "matches", "flush", "tail"]
                dt = ((t[:, 1:] - t[:, :-1]) & 0xffffffff) / 100.0        # us (100 MHz)
                line["phase_us_median"] = {n: round(float(np.median(dt[:, i])), 2) for i, n in enumerate(names)}
                line["block_us_median"] = round(float(np.median(((t[:, 11] - t[:, 0]) & 0xffffffff) / 100.0)), 2)
                t12 = d[ok][:, 12] & 0xffff                              # the match phase: fill | rounds | gather (us)
                f = ((d[ok][:, 3] & 0xffff) - t12) & 0xffff; r = ((d[ok][:, 3] >> 16) - (d[ok][:, 3] & 0xffff)) & 0xffff; g = ((d[ok][:, 13] & 0xffff) - (d[ok][:, 3] >> 16)) & 0xffff
                line["matches_fill_rounds_gather_us"] = [round(float(np.median(x)) / 100.0, 2) for x in (f, r, g)]
                line["shader_MHz"] = round(float(np.median(d[ok][:, 2] / (((t[:, 11] - t[:, 0]) & 0xffffffff) / 100.0))), 0)
            eng.set_decoder(0)
            if "brief" in args:
                ph = line.get("phase_us_median", {})
                print("%-8s %-9s v2 %.4f ms  v4 %.4f ms %6.1f GB/s  block %6.1f us  records %5.2f  matches %5.2f %s  T %.2f cands %.2f rank %.2f lit %.2f" % (
                    kind, writer, line["variant2_ms"], line["variant4_ms"], line["variant4_GBps"], line.get("block_us_median", 0), ph.get("records", 0), ph.get("matches", 0),
                    line.get("matches_fill_rounds_gather_us"), ph.get("T", 0), ph.get("cands", 0), ph.get("rank", 0), ph.get("literals", 0)), flush=True)
            else:
                print(line, flush=True)


if not args or "check" in args:
    check()
if not args or "time" in args:
    timing()
