#!/usr/bin/env python3
"""bench_file.py -- the reference's external-file benchmark protocol (reference benchmark/Main.hs:174-220)
over the GPU engine: BASELINE.json configs[0] ("Canterbury alice29.txt, c+1+65536").

    BENCH_STREAMLY_LZ4_FILE=path  BENCH_STREAMLY_LZ4_STRATEGY=c+<accel>+<bufsize> | d+<bufsize> | r+<bufsize>  python bench_file.py

Like the reference: the file is read in <bufsize> chunks (File.readChunksWithBufferOf), pushed through
compressChunks / decompressChunks / resizeChunks (the C++ mirror of the combinators, GPU codec underneath),
and drained; the timing includes the file read.  The Canterbury corpus is not shipped and cannot be
downloaded here: without BENCH_STREAMLY_LZ4_FILE this prints a message and exits 0 -- nothing is substituted.
With --cpu the same strategy is also timed on the reference codec (oracle/_ref) on one host core.
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))


def parse_strategy(s):                                   # benchmark/Main.hs:195-201
    if s[:1] == "c":
        speed, buf = s[2:].split("+", 1)
        return ("compress", int(speed), int(buf))
    if s[:1] == "d":
        return ("decompress", None, int(s[2:]))
    if s[:1] == "r":
        return ("resize", None, int(s[2:]))
    raise SystemExit("Cannot parse BENCH_STREAMLY_LZ4_STRATEGY")


def read_chunks(path, bufsize):
    with open(path, "rb") as f:
        while True:
            b = f.read(bufsize)
            if not b:
                return
            yield b


def main():
    path = os.environ.get("BENCH_STREAMLY_LZ4_FILE")
    strat = os.environ.get("BENCH_STREAMLY_LZ4_STRATEGY")
    if not path or not strat:
        print("bench_file.py: BENCH_STREAMLY_LZ4_FILE / BENCH_STREAMLY_LZ4_STRATEGY not set "
              "(e.g. corpora/cantrbry/alice29.txt and c+1+65536); the Canterbury corpus is not available offline. Skipping.")
        return
    if not os.path.exists(path):
        print("bench_file.py: %s does not exist. Skipping (no substitute input is used)." % path)
        return
    import streamly_lz4_amd as S
    mode, speed, bufsize = parse_strategy(strat)
    eng = S.Engine(0)
    cfg = S.defaultBlockConfig
    size = os.path.getsize(path)
    reps, best = 5, 1e30
    out_bytes = 0
    for _ in range(reps):
        t0 = time.perf_counter()
        chunks = list(read_chunks(path, bufsize))
        if mode == "compress":
            out = S.compressChunks(cfg, speed, chunks, eng)
        elif mode == "decompress":
            out = S.decompressChunks(cfg, chunks, eng)
        else:
            out = S.resizeChunks(cfg, S.defaultFrameConfig, chunks)
        out_bytes = sum(len(a) for a in out)               # drain
        best = min(best, time.perf_counter() - t0)
    line = {"benchmark": "bufsize(%d)/%s%s/%s" % (bufsize, mode, "" if speed is None else " %d" % speed, path),
            "strategy": strat, "input_bytes": size, "output_bytes": out_bytes, "seconds": round(best, 6),
            "GBps_input": round(size / best / 1e9, 4), "includes_file_read": True, "engine": "mi355lz4 (GPU)"}
    if "--cpu" in sys.argv and mode == "compress":
        from oracle.oracle import Reference
        data = open(path, "rb").read()
        R = Reference()
        t0 = time.perf_counter()
        c = R.frame_compress(data, bufsize, speed, 8, True)
        line["cpu_reference_seconds"] = round(time.perf_counter() - t0, 6)
        line["cpu_reference_output_bytes"] = len(c)
    print(json.dumps(line))


if __name__ == "__main__":
    main()
