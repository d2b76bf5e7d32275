"""Development aid: compressed size against the reference's _continue stream, per input / block size / acceleration,
one-wave-per-block and segmented (the tolerances of tests/test_parity_gpu.py::test_encode_size_vs_reference)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import streamly_lz4_amd as S
from oracle.oracle import Oracle
O = Oracle(); eng = S.Engine(0)
for segs in (0, -1):
    eng.set_segments(segs)
    for kind in ("lzsynth", "text", "random"):
        for bl in (65536, 262144):
            for accel in (1, 5):
                n = 8
                data = O.gen(kind, n, bl).tobytes()
                ours = len(eng.compress_batch([data[i * bl:(i + 1) * bl] for i in range(n)], accel=accel)[0])
                ref = len(O.frame_compress(data, bl, accel, 8, True))
                ind = sum(len(O.compress_block(data[i * bl:(i + 1) * bl], accel)) + 8 for i in range(n))
                print("segs %2d %-8s bl %6d accel %d: ours/ref_linked %.4f  ours/ref_independent %.4f" % (segs, kind, bl, accel, ours / ref, ours / ind))
