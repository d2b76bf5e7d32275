#!/usr/bin/env python3
"""bench_matrix.py -- the reference's gauge benchmark matrix (reference benchmark/Main.hs:226-315) over the GPU
engine's stream combinators (the C++ mirror of compressChunks / decompressChunks / resizeChunks /
decompressChunksWith, GPU codec underneath).

Protocol as in the reference (benchmark/Main.hs:70-121): every input file is normalised to 10 MiB by cycling
it, cut into 64 KiB arrays, and pre-compressed three ways -- "big" (acceleration 65537), "small"
(acceleration 1) and "with" (frame header + blocks + end mark); a benchmark then reads its file in
<bufsize> chunks, runs the combinator and drains the result.  Groups: compress/files (accel 5),
decompress/files/big, decompressWith, decompression/files/small, compression/acceleration {-1, 10, 1000,
65537}, compression/buffer, decompression/buffer, resizing/buffer {6.4 KiB, 64 KiB, 640 KiB}.

Inputs: the Canterbury files the reference uses (large/bible.txt, large/world192.txt, cantrbry/alice29.txt)
under $CANTERBURY_DIR or <repo>/corpora.  They cannot be downloaded here; when they are absent this prints
a message and exits 0 -- unless --synthetic is given, which runs the same matrix on 10 MiB of the engine's
text-like generator and says so in every line.  One JSON line per benchmark; --cpu adds the reference codec
(oracle/_ref) on one host core for the compress / decompress rows.  The decompress rows take the result arrays as
slices of one buffer (views=True), which is what the reference's `Array Word8` results are.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "streamly-lz4_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

_64KB = 64 * 1024
NORMALIZED = 10 * 1024 * 1024                                        # benchmark/Main.hs:80-84
FRAME_HEADER = bytes([0x04, 0x22, 0x4D, 0x18, 0x40, 0x40, 0x00])    # magic, FLG, BD (64 KiB), HC; test/Main.hs:145-151
END_MARK = bytes(4)


def chunks_of(data, n):
    return [data[i:i + n] for i in range(0, len(data), n)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--synthetic", action="store_true", help="run on 10 MiB of the text-like generator when the corpus is absent")
    ap.add_argument("--cpu", action="store_true", help="also time the reference codec on one host core")
    ap.add_argument("--linked", action="store_true",
                    help="linked compression (previous block = dictionary): the pre-compressed inputs are linked streams like "
                         "the reference's own, and so is what the compress benchmarks write")
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    import corpus
    names = ["large/bible.txt", "large/world192.txt", "cantrbry/alice29.txt"]
    files = {n: corpus.find(n) for n in names}
    inputs, label = {}, "canterbury"
    if all(files.values()):
        for n, pth in files.items():
            inputs[n] = corpus.cycled(pth, NORMALIZED)
    elif args.synthetic:
        label = "SYNTHETIC text-like generator (Canterbury corpus absent)"
    else:
        print("bench_matrix.py: Canterbury files %s not found under %s (no network here; set CANTERBURY_DIR, or pass "
              "--synthetic to run the matrix on generated text). Skipping." % ([n for n, p in files.items() if not p], corpus.corpus_dir()))
        return
    import streamly_lz4_amd as S
    eng = S.Engine(0)
    eng.set_linked_compress(args.linked)
    if not inputs:
        import torch
        for k, n in enumerate(names):
            t = torch.empty(NORMALIZED, dtype=torch.uint8, device="cuda:0")
            eng.generate("text", t, _64KB, NORMALIZED // _64KB, first_block=1000 * k)
            eng.synchronize()
            inputs[n] = t.cpu().numpy().tobytes()
    cfg, fcfg = S.defaultBlockConfig, S.defaultFrameConfig
    ref = None
    if args.cpu:
        from oracle.oracle import Reference, have_reference
        ref = Reference() if have_reference() else None

    def compressed(data, accel):
        return b"".join(S.compressChunks(cfg, accel, chunks_of(data, _64KB), eng))

    big = {n: compressed(d, 65537) for n, d in inputs.items()}          # benchmark/Main.hs:85-103
    small = {n: compressed(d, 1) for n, d in inputs.items()}
    withf = {}
    for n, d in inputs.items():
        c64 = S.BlockConfig(S.BlockSize.BlockMax64KB)
        withf[n] = FRAME_HEADER + b"".join(S.compressChunks(c64, 1, chunks_of(d, _64KB), eng)) + END_MARK

    def bench(group, name, data, bufsize, fn, raw_bytes, cpu_fn=None):
        best, out_bytes = 1e30, 0
        for _ in range(args.reps):
            t0 = time.perf_counter()
            out = fn(chunks_of(data, bufsize))
            out_bytes = sum(len(a) for a in out)                            # drain
            best = min(best, time.perf_counter() - t0)
        line = {"group": group, "benchmark": "bufsize(%d)/%s" % (bufsize, name), "input": label,
                "blocks": "linked" if args.linked else "independent", "input_bytes": len(data),
                "output_bytes": out_bytes, "seconds": round(best, 6), "GBps_uncompressed": round(raw_bytes / best / 1e9, 4)}
        if cpu_fn is not None and ref is not None:
            t0 = time.perf_counter()
            cpu_fn()
            line["cpu_reference_seconds_1core"] = round(time.perf_counter() - t0, 6)
        print(json.dumps(line), flush=True)

    for n, d in inputs.items():
        bench("compress/files", "compress 5/" + n, d, _64KB, lambda a: S.compressChunks(cfg, 5, a, eng), len(d),
              (lambda d=d: ref.frame_compress(d, _64KB, 5, 8, True)) if ref else None)
    for n, d in inputs.items():
        bench("decompress/files/big", "decompress/" + n, big[n], _64KB, lambda a: S.decompressChunks(cfg, a, eng, views=True), len(d),
              (lambda n=n, d=d: ref.frame_decompress(big[n], len(d), 8, 0, True)) if ref else None)
    for n, d in inputs.items():
        bench("decompressWith", "decompressWith/" + n, withf[n], _64KB, lambda a: S.decompressChunksWith(a, eng), len(d))
    for n, d in inputs.items():
        bench("decompression/files/small", "decompress/" + n, small[n], _64KB, lambda a: S.decompressChunks(cfg, a, eng, views=True), len(d),
              (lambda n=n, d=d: ref.frame_decompress(small[n], len(d), 8, 0, True)) if ref else None)
    bible = names[0]
    for accel in (-1, 10, 1000, 65537):
        bench("compression/acceleration", "compress %d/%s" % (accel, bible), inputs[bible], _64KB,
              lambda a, accel=accel: S.compressChunks(cfg, accel, a, eng), len(inputs[bible]))
    for buf in (_64KB // 10, _64KB, _64KB * 10):
        bench("compression/buffer", "compress 5/" + bible, inputs[bible], buf, lambda a: S.compressChunks(cfg, 5, a, eng), len(inputs[bible]))
    for buf in (_64KB // 10, _64KB, _64KB * 10):
        bench("decompression/buffer", "decompress/" + bible, big[bible], buf, lambda a: S.decompressChunks(cfg, a, eng, views=True), len(inputs[bible]))
    for buf in (_64KB // 10, _64KB, _64KB * 10):
        bench("resizing/buffer", "resize/" + bible, big[bible], buf, lambda a: S.resizeChunks(cfg, fcfg, a), len(inputs[bible]))


if __name__ == "__main__":
    main()
