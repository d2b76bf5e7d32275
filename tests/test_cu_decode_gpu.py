"""The workgroup-per-block decoder (decoder variant 4, csrc/decode_cu.hpp; what variant 0 picks for calls of up to 256
-- big blocks: 512 -- independent blocks): the same bytes and the same per-block results -- the reference's negative codes included
(cbits/lz4.c:2163) -- as the lane-parallel decoder and the oracle, on the shapes that move it between its forms:
blocks of several segments (a segment is 32 KiB of output or 22 KiB of compressed bytes), lengths with one, two and many
extension bytes (the parse follows two), long literal stretches inside compressible data, matches that overlap their own
output, sources in front of a segment, tiny blocks (left to the lane-parallel decoder), corrupted blocks."""
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _frame_ref(oracle, blocks):
    """independent blocks written by the oracle (= the reference's bytes), 8-byte headers"""
    out = []
    for b in blocks:
        c = oracle.compress_block(b, 1)
        out.append(len(c).to_bytes(4, "little") + len(b).to_bytes(4, "little") + c)
    return b"".join(out)


def _both(engine, framed, expect=None, what=""):
    res = {}
    try:
        for v in (2, 4):
            engine.set_decoder(v)
            res[v] = engine.decompress_batch(framed, raise_on_block_error=False)
    finally:
        engine.set_decoder(0)
    assert res[2][1] == res[4][1], (what, [(i, a, b) for i, (a, b) in enumerate(zip(res[2][1], res[4][1])) if a != b][:8])
    if res[2][0] != res[4][0]:
        a, b = np.frombuffer(res[2][0], np.uint8), np.frombuffer(res[4][0], np.uint8)
        raise AssertionError((what, "bytes differ at", np.nonzero(a != b)[0][:8].tolist()))
    if expect is not None:
        assert res[4][0] == expect, what
    return res[4]


@pytest.mark.parametrize("kind", ["lzsynth", "text", "random"])
def test_cu_decoder_full_blocks(engine, oracle, kind):
    raw = oracle.gen(kind, 24, 65536, first_block=100).tobytes()
    blocks = [raw[i:i + 65536] for i in range(0, len(raw), 65536)]
    _both(engine, _frame_ref(oracle, blocks), raw, kind + " reference-written")
    _both(engine, engine.compress_batch(blocks)[0], raw, kind + " engine-written")
    # the decoder a call of this size gets by default is this one: same answer
    out, blen = engine.decompress_batch(_frame_ref(oracle, blocks))
    assert out == raw and blen == [65536] * 24


def test_cu_decoder_odd_shapes(engine, oracle):
    from test_fuzz_encode_gpu import _make
    rng = random.Random(7)
    blocks = [_make(rng, oracle, t) for t in range(200)]
    text = oracle.gen("text", 1, 30000).tobytes()
    blocks += [bytes(65536), bytes(40000), b"ab" * 30000, bytes(range(256)) * 200, oracle.gen("text", 1, 70000).tobytes(),
               text + oracle.gen("random", 1, 5000).tobytes() + oracle.gen("text", 1, 30000, first_block=9).tobytes(),   # a literal run of 5000 inside text
               text[:9000] + oracle.gen("random", 1, 400).tobytes() + text[9000:],                                            # ... of 400 (two extension bytes)
               text[:20000] + text[1000:1500] + text[20000:],                                                                  # a match of 500
               oracle.gen("lzsynth", 1, 262144).tobytes(), oracle.gen("text", 1, 1 << 20, first_block=3).tobytes(),
               oracle.gen("random", 1, 100000).tobytes(), oracle.gen("text", 1, 1000).tobytes(), b"", b"x",
               oracle.gen("text", 1, 65535).tobytes(), (oracle.gen("text", 1, 3000).tobytes() + bytes(1500)) * 14]
    raw = b"".join(blocks)
    _both(engine, _frame_ref(oracle, blocks), raw, "odd shapes, reference-written")
    _both(engine, engine.compress_batch(blocks)[0], raw, "odd shapes, engine-written")


def test_cu_decoder_big_blocks(engine, oracle):
    """BlockMax1MB / BlockMax4MB sized blocks (Config.hs:109-116): 32 and 128 segments a block, sources in earlier segments."""
    for n, kind in ((1 << 20, "text"), (4 << 20, "lzsynth"), ((4 << 20) - 7, "text")):
        data = oracle.gen(kind, (n + 65535) // 65536, 65536, first_block=11).tobytes()[:n]
        # text repeats itself across the block: matches reach back up to 65535 bytes, into earlier segments
        fr = _frame_ref(oracle, [data, data[: n // 3]])
        _both(engine, fr, data + data[: n // 3], "big blocks %d %s" % (n, kind))


def test_cu_decoder_corrupted_blocks(engine, oracle):
    rng = random.Random(5)
    base = [oracle.gen("text", 1, 65536, first_block=5).tobytes(), oracle.gen("lzsynth", 1, 65536, first_block=6).tobytes(),
            oracle.gen("text", 1, 200000, first_block=8).tobytes()]
    nbad = 0
    for trial in range(240):
        b = base[trial % 3]
        c = bytearray(oracle.compress_block(b, 1))
        for _ in range(rng.choice((1, 1, 2, 5))):
            pos = rng.randrange(len(c)) if trial % 4 else rng.randrange(max(1, len(c) - 200), len(c))
            c[pos] = rng.choice((rng.randrange(256), 0xFF, 0x00))
        cl = len(c) if trial % 7 else len(c) - rng.randrange(1, 40)
        ul = len(b) if trial % 5 else len(b) - rng.randrange(0, 300)
        fr = cl.to_bytes(4, "little") + ul.to_bytes(4, "little") + bytes(c[:cl])
        out, blen = _both(engine, fr, None, "corrupted %d" % trial)
        code, want = oracle.decompress_block(bytes(c[:cl]), ul)
        assert blen == [code], (trial, blen, code)
        if code >= 0:
            assert out == want
        nbad += code < 0
    assert nbad > 40


def test_cu_decoder_is_the_default_for_small_calls_only(engine, oracle):
    """Variant 0 (api.cpp, cu_auto): calls of up to 256 blocks -- 512 when the blocks hold 16 KiB of compressed bytes or more on
    average, none when less than 3 KiB -- go to the workgroup-per-block decoder, the others to the lane-parallel one (the 16 words
    of diagnostics per block tell which ran); a linked call's first -- standalone -- pass follows the same rule; and a block that
    saves less than a sixteenth of its size is handed on by the kernel itself (long literal runs end that form's segments)."""
    import ctypes as C
    import torch
    S = pytest.importorskip("streamly_lz4_amd")
    dev = torch.device("cuda:0")
    cases = ((100, 16384, False, True), (256, 16384, False, True), (257, 16384, False, False), (100, 16384, True, True),
             (300, 16384, True, False), (512, 65536, False, True), (513, 65536, False, False), (100, 4096, False, False),
             (100, -65536, False, False))            # (blocks that hardly compress -- here: not at all -- are left to the lane-parallel decoder by the kernel)
    for nblk, bl, linked, expect in cases:
        kind = "text" if bl > 0 else "random"
        bl = abs(bl)
        raw = oracle.gen(kind, nblk, bl, first_block=1).tobytes()
        blocks = [raw[i:i + bl] for i in range(0, len(raw), bl)]
        fr = _frame_ref(oracle, blocks)
        offs, pos = [], 0
        for _ in range(nblk):
            offs.append(pos)
            pos += 8 + int.from_bytes(fr[pos:pos + 4], "little")
        buf = torch.frombuffer(bytearray(fr), dtype=torch.uint8).to(dev)
        boff = torch.tensor(offs + [pos], dtype=torch.int64, device=dev)
        ooff = torch.arange(0, (nblk + 1) * bl, bl, dtype=torch.int64, device=dev)
        out = torch.zeros(nblk * bl, dtype=torch.uint8, device=dev)
        res = torch.zeros(nblk, dtype=torch.int32, device=dev)
        dbg = torch.zeros(nblk * 16, dtype=torch.int32, device=dev)
        S.lib.mi355lz4_debug_cu(engine.ctx, C.c_void_p(dbg.data_ptr()))
        try:
            engine.decompress_batch_device(buf, len(fr), boff, nblk, out, ooff, res, linked=linked)
            engine.synchronize()
        finally:
            S.lib.mi355lz4_debug_cu(engine.ctx, None)
        assert out.cpu().numpy().tobytes() == raw and bool((res == bl).all())
        ran = bool((dbg.view(nblk, 16)[:, 15] != 0).any().item())       # [15]: the clock at the block's end
        assert ran == expect, (nblk, bl, linked, ran)


def test_runin_state_decays_with_every_linked_call(engine, oracle):
    """The run-in decode's adaptive state (api.cpp: runinLong, runinLongOk, runinSkip): an engine that was sent to the long run-in
    tries the default again after RUNIN_LONG_PROBE (32) linked calls of ANY size, not only after that many long run-ins."""
    import ctypes as C
    S = pytest.importorskip("streamly_lz4_amd")
    f = S.lib.mi355lz4_debug_runin_state
    f.restype = C.c_int
    st = (C.c_int * 5)()
    data = oracle.gen("text", 8, 65536).tobytes()
    fr = oracle.frame_compress(data, 65536, 1, 8, True)                 # a linked stream far below every run-in threshold
    try:
        assert f(engine.ctx, None, (C.c_int * 3)(1, 0, 0)) == 0         # as if a call had been given up with the default run-in
        for k in range(31):
            out, _ = engine.decompress_batch(fr, linked=True)
            assert out == data
            f(engine.ctx, st, None)
            assert list(st)[:3] == [1, k + 1, 0], (k, list(st))
        out, _ = engine.decompress_batch(fr, linked=True)
        f(engine.ctx, st, None)
        assert out == data and list(st)[:3] == [0, 0, 0]
    finally:
        f(engine.ctx, None, (C.c_int * 3)(0, 0, 0))


def _linked_device_call(S, engine, fr, nblk, bl, raw_len):
    """(bytes, results, path) of one device-resident linked call over a framed stream of nblk blocks of bl bytes (the last may be short)"""
    import ctypes as C
    import torch
    dev = torch.device("cuda:0")
    offs, pos = [], 0
    for _ in range(nblk):
        offs.append(pos)
        pos += 8 + int.from_bytes(fr[pos:pos + 4], "little")
    buf = torch.frombuffer(bytearray(fr), dtype=torch.uint8).to(dev)
    boff = torch.tensor(offs + [pos], dtype=torch.int64, device=dev)
    ooff = torch.arange(0, (nblk + 1) * bl, bl, dtype=torch.int64, device=dev)
    out = torch.zeros(nblk * bl, dtype=torch.uint8, device=dev)
    res = torch.zeros(nblk, dtype=torch.int32, device=dev)
    engine.decompress_batch_device(buf, len(fr), boff, nblk, out, ooff, res, linked=True)
    engine.synchronize()
    st = (C.c_int * 5)()
    S.lib.mi355lz4_debug_runin_state(engine.ctx, st, None)
    return out.cpu().numpy().tobytes()[:raw_len], res.cpu().tolist(), st[4]


def test_big_linked_blocks_by_the_workgroup_form(engine, oracle, monkeypatch):
    """A linked stream of BlockMax1MB-sized blocks (Config.hs:109-116; written with the dictionary carried from block to block,
    cbits/lz4.c:1608-1636): every dependent block by the workgroup-per-block decoder against a guess of its dictionary, pass after
    pass until the guesses stand (api.cpp path 6, kernels.hip k_decode_cu_linked).  Bytes and results are the input's and those of
    the pointer pass (MI355LZ4_LINKED_BIG=0); a corrupted block, and a stream whose blocks never forget their dictionary, leave the
    call to that pass with the same results."""
    S = pytest.importorskip("streamly_lz4_amd")
    monkeypatch.delenv("MI355LZ4_LINKED_BIG", raising=False)
    for bl, nblk, kind in ((1 << 20, 24, "text"), (4 << 20, 5, "text"), (512 << 10, 40, "text")):
        raw = oracle.gen(kind, nblk * bl // 65536, 65536, first_block=21).tobytes()[: nblk * bl - 777]     # (a ragged last block)
        fr = oracle.frame_compress(raw, bl, 1, 8, True)
        out, res, path = _linked_device_call(S, engine, fr, nblk, bl, len(raw))
        assert path == 6, (bl, path)
        assert out == raw and res == [bl] * (nblk - 1) + [bl - 777]
        monkeypatch.setenv("MI355LZ4_LINKED_BIG", "0")
        out0, res0, path0 = _linked_device_call(S, engine, fr, nblk, bl, len(raw))
        monkeypatch.delenv("MI355LZ4_LINKED_BIG")
        assert path0 != 6 and out0 == raw and res0 == res
        # a corrupted block in the middle: the same results either way (the codes are the exact path's)
        bad = bytearray(fr)
        pos = 0
        for _ in range(nblk // 2):
            pos += 8 + int.from_bytes(fr[pos:pos + 4], "little")
        clen = int.from_bytes(fr[pos:pos + 4], "little")
        bad[pos + 8 + clen // 2: pos + 8 + clen // 2 + 40] = b"\xff" * 40        # (length fields that run on and on)
        bad[pos + 8 + 100: pos + 8 + 104] = b"\x0f\x00\x00\x00"                # (and a match with offset 0 near the block's start)
        o1, r1, p1 = _linked_device_call(S, engine, bytes(bad), nblk, bl, len(raw))
        monkeypatch.setenv("MI355LZ4_LINKED_BIG", "0")
        o0, r0, p0 = _linked_device_call(S, engine, bytes(bad), nblk, bl, len(raw))
        monkeypatch.delenv("MI355LZ4_LINKED_BIG")
        assert r1 == r0, (bl, p1, [(i, a, b) for i, (a, b) in enumerate(zip(r1, r0)) if a != b])
        assert p1 != 6 or all(x >= 65536 for x in r1[:-1]), (bl, p1)    # (a failing block sends the call to the exact passes)
        good = [i for i, x in enumerate(r0) if x > 0 and all(y > 0 for y in r0[:i + 1])]
        assert all(o1[i * bl:(i + 1) * bl] == o0[i * bl:(i + 1) * bl] for i in good)
    # a failure in the LAST block (nobody's dictionary): the call must not be finished by this path with that block's result
    bl, nblk = 512 << 10, 3
    raw = oracle.gen("text", nblk * bl // 65536, 65536, first_block=5).tobytes()
    fr = bytearray(oracle.frame_compress(raw, bl, 1, 8, True))
    pos = 0
    for _ in range(nblk - 1):
        pos += 8 + int.from_bytes(fr[pos:pos + 4], "little")
    clen = int.from_bytes(fr[pos:pos + 4], "little")
    fr[pos + 8 + clen - 30: pos + 8 + clen] = b"\xff" * 30         # (the block's end rules cannot hold: cbits/lz4.c:1929-2151)
    o1, r1, p1 = _linked_device_call(S, engine, bytes(fr), nblk, bl, len(raw))
    monkeypatch.setenv("MI355LZ4_LINKED_BIG", "0")
    o0, r0, p0 = _linked_device_call(S, engine, bytes(fr), nblk, bl, len(raw))
    monkeypatch.delenv("MI355LZ4_LINKED_BIG")
    assert r1 == r0 and r1[-1] < 0 and p1 != 6 and o1[: 2 * bl] == raw[: 2 * bl], (r1, r0, p1)
    # a short block in the middle of big ones (resized chunks), written by the engine's linked compressor: the block behind it leans
    # on a dictionary that is no 64 KiB of one block -- not this path's; bytes and results must be right all the same
    parts = [oracle.gen("text", 16, 65536, first_block=31).tobytes(), oracle.gen("text", 16, 65536, first_block=31).tobytes()[500000:530000],
             oracle.gen("text", 16, 65536, first_block=31).tobytes()[100000:1000000]]
    e2 = S.Engine(0)
    try:
        e2.set_linked_compress(True)
        fr3, flens = e2.compress_batch(parts)
    finally:
        e2.close()
    import torch
    dev = torch.device("cuda:0")
    cap = 1 << 20
    offs = [0, flens[0], flens[0] + flens[1], len(fr3)]
    buf = torch.frombuffer(bytearray(fr3), dtype=torch.uint8).to(dev)
    boff = torch.tensor(offs, dtype=torch.int64, device=dev)
    ooff = torch.tensor([0, cap, 2 * cap, 3 * cap], dtype=torch.int64, device=dev)
    out = torch.zeros(3 * cap, dtype=torch.uint8, device=dev)
    res = torch.zeros(3, dtype=torch.int32, device=dev)
    engine.decompress_batch_device(buf, len(fr3), boff, 3, out, ooff, res, linked=True)
    engine.synchronize()
    got = out.cpu().numpy().tobytes()
    assert res.cpu().tolist() == [len(x) for x in parts]
    assert all(got[i * cap:i * cap + len(x)] == x for i, x in enumerate(parts))
    # the host-buffer call: groups of 64 MiB; the second group's first block leans on the first group's last, which is final by then
    import ctypes as C
    bl, nblk = 1 << 20, 72
    raw = oracle.gen("text", nblk * bl // 65536, 65536, first_block=41).tobytes()[: nblk * bl - 4321]
    fr = oracle.frame_compress(raw, bl, 1, 8, True)
    out, blen = engine.decompress_batch(fr, linked=True)
    st = (C.c_int * 5)()
    S.lib.mi355lz4_debug_runin_state(engine.ctx, st, None)
    assert out == raw and blen == [bl] * (nblk - 1) + [bl - 4321] and st[4] == 6, st[4]
    # blocks that never forget: every 1 MiB block is the block before it, shifted (one long match out of the dictionary, then itself)
    import random
    bl, nblk = 1 << 20, 12
    pat = random.Random(3).randbytes(65000)
    raw = (pat * (nblk * bl // len(pat) + 1))[: nblk * bl]
    fr = oracle.frame_compress(raw, bl, 1, 8, True)
    out, res, path = _linked_device_call(S, engine, fr, nblk, bl, len(raw))
    assert out == raw and res == [bl] * nblk, path
