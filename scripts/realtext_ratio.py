"""Development aid: compressed size on REAL text-like files present in the image (Python standard-library sources,
this repo's own sources) against the reference codec, independent blocks and the reference's linked stream; the
Canterbury corpus is not in the image.  One wave per block (segments off), segmented, and linked compression."""
import glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import streamly_lz4_amd as S
from oracle.oracle import Oracle
O = Oracle(); eng = S.Engine(0); BL = 65536
sets = {"python stdlib sources": sorted(glob.glob("/usr/lib/python3.10/*.py")),
        "this repo (C++/HIP/Python/markdown)": sorted(glob.glob(ROOT + "/streamly-lz4_amd/csrc/*") + glob.glob(ROOT + "/*.md") + glob.glob(ROOT + "/tests/*.py"))}
for name, files in sets.items():
    data = b"".join(open(f, "rb").read() for f in files)
    data = data[: len(data) // BL * BL]
    if len(data) < BL:
        continue
    blocks = [data[i:i + BL] for i in range(0, len(data), BL)]
    ref_linked = len(O.frame_compress(data, BL, 1, 8, True))
    ref_indep = sum(len(O.compress_block(b, 1)) + 8 for b in blocks)
    row = {"input": name, "MiB": round(len(data) / 2 ** 20, 2), "reference_linked_ratio": round(len(data) / ref_linked, 4),
           "reference_independent_ratio": round(len(data) / ref_indep, 4)}
    for label, segs, linked in (("one_wave_per_block", 0, False), ("segmented", -1, False), ("linked_compression", 0, True)):
        eng.set_segments(segs); eng.set_linked_compress(linked)
        fr, _ = eng.compress_batch(blocks, accel=1)
        assert O.frame_decompress(fr, len(data), 8, 0, True) == data
        row[label] = {"ratio": round(len(data) / len(fr), 4), "vs_reference_linked": round(len(fr) / ref_linked, 4),
                      "vs_reference_independent": round(len(fr) / ref_indep, 4)}
    eng.set_segments(-1); eng.set_linked_compress(False)
    print(row, flush=True)
