"""Generate tests/golden/*.json from the REAL reference codec (oracle/_ref, built from
/root/reference/cbits/lz4.c by oracle/Makefile).  Run in the build container:

    python tests/golden/make_golden.py

The reference's own tests hold no known-answer vectors (they are all
round-trip properties), so these vectors are outputs of the reference itself run
here.  Fixtures are data only: inputs (or their seeds) and expected outputs.
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.oracle import Oracle, Reference  # noqa: E402

O, R = Oracle(), Reference()


def sha(b):
    return hashlib.sha256(b).hexdigest()


def main():
    assert R.version() == 10903, R.version()
    gold = {"lz4_version": R.version()}

    # (1) known-answer vectors: tiny inputs, full compressed bytes
    ramp = bytes(((i * 7 + (i >> 8)) & 255) for i in range(65536))
    kat = []
    for name, data in [("empty", b""), ("a12", b"a" * 12), ("a13", b"a" * 13), ("a64", b"a" * 64),
                       ("abc_x40", b"abc" * 40), ("hello", b"hello hello hello hello hello"),
                       ("bytes0_255", bytes(range(256)))]:
        kat.append({"name": name, "input_hex": data.hex(), "accel": 1, "compressed_hex": R.compress_block(data, 1).hex()})
    gold["kat"] = kat
    sizes = []
    for name, data in [("zeros64k", bytes(65536)), ("zeros256k", bytes(262144)), ("ramp64k", ramp)]:
        for accel in (-1, 0, 1, 5, 400, 65537, 1000000):
            c = R.compress_block(data, accel)
            sizes.append({"name": name, "accel": accel, "size": len(c), "sha256": sha(c)})
    gold["kat_sizes"] = sizes

    # (2) malformed inputs: (payload, capacity, dict?) -> reference return code
    mal = []
    base = R.compress_block(O.gen("text", 1, 300).tobytes(), 1)
    big = R.compress_block(O.gen("lzsynth", 1, 5000).tobytes(), 1)
    cases = [
        ("trunc_half", base[: len(base) // 2], 300),
        ("trunc_1", base[:-1], 300),
        ("extra_byte", base + b"\x00", 300),
        ("cap_small", base, 299),
        ("cap_big", base, 301),
        ("cap_zero", base, 0),
        ("empty_src_cap0", b"", 0),
        ("zero_token_cap0", b"\x00", 0),
        ("offset_beyond_start", bytes([0x10, 0x41, 0x10, 0x00, 0x50]) + b"abcde", 64),
        ("offset_zero", bytes([0x14, 0x41, 0x00, 0x00, 0x50]) + b"abcde", 64),
        ("ends_in_match", bytes([0x10, 0x41, 0x01, 0x00]), 64),
        ("litlen_overrun", bytes([0xF0, 0xFF, 0xFF]), 1000),
        ("matchlen_overrun", bytes([0x1F, 0x41, 0x01, 0x00, 0xFF, 0xFF]), 5000),
        ("big_trunc", big[:777], 5000),
        ("big_cap_small", big, 4000),
        ("garbage", bytes((i * 37 + 11) & 255 for i in range(200)), 1000),
    ]
    for name, payload, cap in cases:
        code, out = R.decompress_block(payload, cap)
        mal.append({"name": name, "payload_hex": payload.hex(), "cap": cap, "code": code,
                    "out_sha256": sha(out) if code >= 0 else None})
    gold["malformed"] = mal

    # (3) seeded streams: per-block compressed sizes + digest of the framed stream, reference semantics
    streams = []
    for kind in ("lzsynth", "random", "text"):
        for bl in (65536, 262144):
            data = O.gen(kind, 4, bl).tobytes()
            for accel in (1, 5, 400):
                for linked in (True, False):
                    fr = R.frame_compress(data, bl, accel, 8, linked)
                    pos, csz = 0, []
                    while pos < len(fr):
                        c = int.from_bytes(fr[pos:pos + 4], "little")
                        csz.append(c)
                        pos += 8 + c
                    streams.append({"kind": kind, "block_len": bl, "n_blocks": 4, "accel": accel, "linked": linked,
                                    "comp_sizes": csz, "framed_sha256": sha(fr), "raw_sha256": sha(data)})
    gold["streams"] = streams

    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(gold, f, indent=1)

    # (4) a linked 4-block stream (shared vocabulary): blocks 1-3 fail standalone, decode through _continue
    bl = 4096
    data = O.gen("text", 4, bl).tobytes()
    fr = R.frame_compress(data, bl, 1, 8, True)
    blocks, pos = [], 0
    while pos < len(fr):
        c = int.from_bytes(fr[pos:pos + 4], "little")
        blocks.append(fr[pos + 8:pos + 8 + c])
        pos += 8 + c
    standalone = [R.decompress_block(b, bl)[0] for b in blocks]
    assert standalone[0] == bl and all(s < 0 for s in standalone[1:]), standalone
    linked = {"block_len": bl, "framed_hex": fr.hex(), "raw_sha256": sha(data), "standalone_codes": standalone,
              "generator": {"kind": "text", "n_blocks": 4, "first_block": 0}}
    with open(os.path.join(HERE, "linked_stream.json"), "w") as f:
        json.dump(linked, f, indent=1)
    print("wrote golden.json (%d kat, %d sizes, %d malformed, %d streams) and linked_stream.json (%d bytes framed)"
          % (len(kat), len(sizes), len(mal), len(streams), len(fr)))


if __name__ == "__main__":
    main()
