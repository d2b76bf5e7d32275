"""CPU: the oracle restatement against the committed golden vectors (generated from the real
reference, tests/golden/make_golden.py) and, where oracle/_ref exists, against the reference itself."""
import hashlib
import random

import pytest


def sha(b):
    return hashlib.sha256(b).hexdigest()


def test_golden_version(golden):
    assert golden["lz4_version"] == 10903  # cbits/lz4.h:101-103


def test_kat_compressed_bytes(oracle, golden):
    for k in golden["kat"]:
        data = bytes.fromhex(k["input_hex"])
        comp = oracle.compress_block(data, k["accel"])
        assert comp.hex() == k["compressed_hex"], k["name"]
        code, out = oracle.decompress_block(comp, len(data))
        assert code == len(data) and out == data


def test_survey_kats(oracle):
    # SURVEY.md 8c (1)
    assert oracle.compress_block(b"").hex() == "00"
    assert oracle.compress_block(b"a" * 13).hex() == "13610100506161616161"
    assert oracle.compress_block(b"a" * 64).hex() == "1f61010027506161616161"
    assert len(oracle.compress_block(b"a" * 12)) == 13
    assert len(oracle.compress_block(bytes(65536))) == 267
    assert len(oracle.compress_block(bytes(262144))) == 1038


def test_kat_sizes_and_accel_clamp(oracle, golden):
    inputs = {"zeros64k": bytes(65536), "zeros256k": bytes(262144),
              "ramp64k": bytes(((i * 7 + (i >> 8)) & 255) for i in range(65536))}
    for k in golden["kat_sizes"]:
        c = oracle.compress_block(inputs[k["name"]], k["accel"])
        assert len(c) == k["size"] and sha(c) == k["sha256"], k
    by = {(k["name"], k["accel"]): k["size"] for k in golden["kat_sizes"]}
    assert by[("ramp64k", -1)] == by[("ramp64k", 0)] == by[("ramp64k", 1)] == 2249   # cbits/lz4.c:1577
    assert by[("ramp64k", 65537)] == by[("ramp64k", 1000000)] == 65794               # cbits/lz4.c:1578


def test_malformed_codes(oracle, golden):
    for m in golden["malformed"]:
        code, out = oracle.decompress_block(bytes.fromhex(m["payload_hex"]), m["cap"])
        assert code == m["code"], m["name"]
        if code >= 0:
            assert sha(out) == m["out_sha256"], m["name"]


def test_seeded_streams(oracle, golden):
    for s in golden["streams"]:
        data = oracle.gen(s["kind"], s["n_blocks"], s["block_len"]).tobytes()
        assert sha(data) == s["raw_sha256"], ("generator", s["kind"])
        fr = oracle.frame_compress(data, s["block_len"], s["accel"], 8, s["linked"])
        assert sha(fr) == s["framed_sha256"], s
        assert oracle.frame_decompress(fr, len(data), 8, 0, True) == data


def test_linked_stream_fixture(oracle, linked_golden):
    fr = bytes.fromhex(linked_golden["framed_hex"])
    bl = linked_golden["block_len"]
    blocks, pos = [], 0
    while pos < len(fr):
        c = int.from_bytes(fr[pos:pos + 4], "little")
        blocks.append(fr[pos + 8:pos + 8 + c])
        pos += 8 + c
    assert [oracle.decompress_block(b, bl)[0] for b in blocks] == linked_golden["standalone_codes"]
    out = oracle.frame_decompress(fr, 4 * bl, 8, 0, True)
    assert sha(out) == linked_golden["raw_sha256"]
    with pytest.raises(RuntimeError):
        oracle.frame_decompress(fr, 4 * bl, 8, 0, False)   # inter-block dependency is real


def test_compress_bound(oracle):
    assert oracle.compress_bound(65536) == 65809 and oracle.compress_bound(262144) == 263188
    assert oracle.compress_bound(0) == 16 and oracle.compress_bound(0x7E000001) == 0


# ---- against the reference itself (only where oracle/_ref was built) --------------------------
def test_vs_reference_compress_bytes(oracle, reference):
    for kind in ("lzsynth", "text", "random"):
        for bl in (1, 12, 13, 100, 4096, 65536, 70000):
            data = oracle.gen(kind, 3, bl, first_block=11).tobytes()
            for accel in (1, 7, 400):
                for linked in (True, False):
                    for hdr in (8, 4):
                        assert oracle.frame_compress(data, bl, accel, hdr, linked) == \
                            reference.frame_compress(data, bl, accel, hdr, linked), (kind, bl, accel, linked, hdr)


def test_vs_reference_decode_fuzz(oracle, reference):
    rng = random.Random(99)
    for it in range(400):
        kind = rng.choice(["lzsynth", "text", "random"])
        n = rng.choice([0, 1, 12, 13, 20, 64, 65, 100, 300, 2000])
        data = oracle.gen(kind, 1, max(n, 1), first_block=it)[:n].tobytes()
        comp = oracle.compress_block(data, rng.choice([1, 1, 9]))
        probes = [(comp, n), (comp, n + 7), (comp, max(n - 1, 0)), (comp[: max(1, len(comp) // 2)], n)]
        m = bytearray(comp)
        m[rng.randrange(len(m))] = rng.randrange(256)
        probes += [(bytes(m), n), (bytes(m), n + 64)]
        d = oracle.gen("text", 1, rng.choice([3, 500, 65536, 70000]), first_block=it + 5).tobytes()
        for payload, cap in probes:
            for dct in (None, d):
                a = oracle.decompress_block(payload, cap, dct)
                b = reference.decompress_block(payload, cap, dct)
                assert a == b, (it, len(payload), cap, dct is not None, a[0], b[0])


def _huge_length_cases():
    """Blocks whose literal / match length fields are multi-megabyte runs of 0xFF: the lengths reach
    2^31 and beyond (cbits/lz4.c:1811-1818, 1854-1858, 2064-2065 compare them as size_t)."""
    run = 8_600_000                                     # 255 * run > 2^31
    lit = b"\xf0" + b"\xff" * run + b"\x07" + b"abc"    # literal length overflowing int
    mat = b"\x1f" + b"x" + b"\x01\x00" + b"\xff" * run + b"\x03" + b"\x50hello"
    mat_dict = b"\x1f" + b"x" + b"\x05\x00" + b"\xff" * run + b"\x03" + b"\x50hello"
    return [("lit", lit, 4096), ("lit_bigcap", lit, 1 << 20), ("match", mat, 4096), ("match_cap64", mat, 40),
            ("match_far", mat_dict, 1 << 16)]


def test_huge_length_fields_vs_reference(oracle, reference):
    for name, payload, cap in _huge_length_cases():
        assert oracle.decompress_block(payload, cap) == reference.decompress_block(payload, cap), name
        d = bytes(range(256)) * 16
        assert oracle.decompress_block(payload, cap, d) == reference.decompress_block(payload, cap, d), name
