"""Standard LZ4 frame format over the GPU engine: interop, both directions, with the system
liblz4's LZ4F_* API -- an implementation that shares no code with this repo.  The reference's own frame parser
rejects these frames (src/Streamly/Internal/LZ4.hs:631-638); they are what `lz4` writes by default."""
import os
import struct
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(__file__))
import lz4f  # noqa: E402

pytestmark = pytest.mark.gpu

L = lz4f.load()
needs_lz4f = pytest.mark.skipif(L is None, reason="system liblz4 with LZ4F_* not found")


@pytest.fixture(scope="module")
def eng():
    import streamly_lz4_amd as slz
    e = slz.Engine(0)
    yield e
    e.close() if hasattr(e, "close") else None


def _text(n, seed):
    rng = np.random.default_rng(seed)
    words = [bytes(rng.integers(97, 123, size=int(k), dtype=np.uint8)) for k in rng.integers(2, 10, size=600)]
    idx = rng.zipf(1.3, size=n // 4 + 16) % len(words)
    return b" ".join(words[i] for i in idx)[:n]


def _mixed(n, seed):
    """compressible text with incompressible stretches: some blocks get stored"""
    rng = np.random.default_rng(seed)
    out = bytearray()
    while len(out) < n:
        out += _text(int(rng.integers(30000, 150000)), int(rng.integers(1 << 30)))
        out += rng.integers(0, 256, size=int(rng.integers(70000, 140000)), dtype=np.uint8).tobytes()
    return bytes(out[:n])


BLOCK_IDS = {4: 1, 5: 2, 6: 3, 7: 4}      # LZ4F blockSizeID -> BlockSize kind


@needs_lz4f
@pytest.mark.parametrize("size", [0, 1, 13, 65535, 65536, 65537, 1 << 20, 3 * (1 << 20) + 12345])
@pytest.mark.parametrize("opts", [dict(), dict(blockChecksum=True), dict(contentSize=True, contentChecksum=False),
                                  dict(blockChecksum=True, contentSize=True)])
def test_frames_written_here_are_read_by_liblz4(eng, size, opts):
    import streamly_lz4_amd as slz
    data = _mixed(size, 7 + size % 1000)
    frame = slz.lz4FrameCompress(data, eng, **opts)
    assert lz4f.decompress(L, frame, len(data) + 16) == data
    assert slz.lz4FrameDecompress(frame, eng) == data


@needs_lz4f
@pytest.mark.parametrize("block_id", [4, 5, 6, 7])
def test_block_sizes_both_ways(eng, block_id):
    import streamly_lz4_amd as slz
    data = _mixed(5 * (1 << 20) + 777, block_id)
    frame = slz.lz4FrameCompress(data, eng, blockMax=BLOCK_IDS[block_id], blockChecksum=True)
    assert frame[5] >> 4 == block_id
    assert lz4f.decompress(L, frame, len(data) + 16) == data
    theirs = lz4f.compress_frame(L, data, block_id=block_id, linked=False, content_checksum=True, block_checksum=True)
    assert slz.lz4FrameDecompress(theirs, eng) == data


@needs_lz4f
@pytest.mark.parametrize("block_id", [4, 5, 7])
def test_linked_frames_written_here(eng, block_id):
    """Block-dependent frames (what `lz4` writes by default), written with linked compression: liblz4 reads them,
    so does this reader, and on text they are smaller than the independent-block frame of the same data."""
    import streamly_lz4_amd as slz
    for size in (0, 70000, 3 * (1 << 20) + 4321):
        data = _text(size, 40 + block_id)
        frame = slz.lz4FrameCompress(data, eng, blockMax=BLOCK_IDS[block_id], linkedBlocks=True, blockChecksum=True)
        assert len(frame) >= 7 and not frame[4] & 0x20
        assert lz4f.decompress(L, frame, len(data) + 16) == data
        assert slz.lz4FrameDecompress(frame, eng) == data
        if size > (1 << 20) and block_id == 4:
            indep = slz.lz4FrameCompress(data, eng, blockMax=BLOCK_IDS[block_id], blockChecksum=True)
            assert len(frame) < 0.99 * len(indep)
    mixed = _mixed(2 * (1 << 20), 77)                       # stored blocks inside a linked frame
    frame = slz.lz4FrameCompress(mixed, eng, linkedBlocks=True)
    assert lz4f.decompress(L, frame, len(mixed) + 16) == mixed and slz.lz4FrameDecompress(frame, eng) == mixed


@needs_lz4f
@pytest.mark.parametrize("linked", [False, True])
@pytest.mark.parametrize("size", [0, 5, 65536, 200000, 4 * (1 << 20) + 99])
@pytest.mark.parametrize("kw", [dict(), dict(content_checksum=True, block_checksum=True), dict(level=9)])
def test_frames_written_by_liblz4_are_read_here(eng, linked, size, kw):
    import streamly_lz4_amd as slz
    data = _mixed(size, 100 + size % 977) if size > 65536 else _text(size, 3)
    kw = dict(kw)
    if size:
        kw["content_size"] = size
    frame = lz4f.compress_frame(L, data, block_id=4, linked=linked, **kw)
    assert slz.lz4FrameDecompress(frame, eng) == data


@needs_lz4f
def test_linked_frame_with_short_blocks_in_mid_frame(eng):
    """A streaming writer that flushes: the window of a block spans several short blocks before it."""
    import streamly_lz4_amd as slz
    rng = np.random.default_rng(11)
    base = _text(1 << 20, 5)
    pieces, pos = [], 0
    while pos < len(base):
        n = int(rng.integers(100, 90000))
        pieces.append(base[pos:pos + n])
        pos += n
    frame = lz4f.compress_pieces(L, pieces, block_id=4, linked=True, content_checksum=True)
    assert lz4f.decompress(L, frame, len(base) + 16) == base
    assert slz.lz4FrameDecompress(frame, eng) == base
    frame = lz4f.compress_pieces(L, pieces, block_id=5, linked=False)
    assert slz.lz4FrameDecompress(frame, eng) == base


@needs_lz4f
def test_concatenated_and_skippable_frames(eng):
    import streamly_lz4_amd as slz
    a, b = _text(100000, 1), _mixed(300000, 2)
    skip = struct.pack("<II", 0x184D2A53, 11) + b"hello world"
    stream = lz4f.compress_frame(L, a, linked=True) + skip + slz.lz4FrameCompress(b, eng) + skip
    assert slz.lz4FrameDecompress(stream, eng) == a + b
    assert slz.lz4FrameDecompress(b"", eng) == b""


def test_stored_blocks_are_flagged(eng):
    import streamly_lz4_amd as slz
    data = np.random.default_rng(3).integers(0, 256, size=200000, dtype=np.uint8).tobytes()
    frame = slz.lz4FrameCompress(data, eng, contentChecksum=False)
    (word,) = struct.unpack_from("<I", frame, 7)
    assert word == 0x80000000 | 65536
    assert len(frame) == 7 + 4 * 4 + len(data) + 4          # 4 block words, end mark
    assert slz.lz4FrameDecompress(frame, eng) == data


@needs_lz4f
def test_corruption_is_detected(eng):
    import streamly_lz4_amd as slz
    data = _text(300000, 9)
    frame = bytearray(lz4f.compress_frame(L, data, linked=True, content_checksum=True, block_checksum=True, content_size=len(data)))

    def bad(mut, msg):
        f = bytearray(frame)
        mut(f)
        with pytest.raises(slz.LZ4Error, match=msg):
            slz.lz4FrameDecompress(bytes(f), eng)

    bad(lambda f: f.__setitem__(0, 5), "bad magic")
    bad(lambda f: f.__setitem__(4, f[4] ^ 0x08), "header checksum|truncated")
    bad(lambda f: f.__setitem__(14, f[14] ^ 1), "header checksum")
    bad(lambda f: f.__setitem__(40, f[40] ^ 0x55), "block checksum")
    bad(lambda f: f.__setitem__(len(f) - 1, f[-1] ^ 1), "content checksum")
    bad(lambda f: f.__delitem__(slice(len(f) - 9, len(f))), "truncated")
    # without block checksums a damaged block is caught by the decoder or by the content checksum
    frame = bytearray(lz4f.compress_frame(L, data, linked=False, content_checksum=True))
    f = bytearray(frame)
    f[5000] ^= 0xFF
    with pytest.raises(slz.LZ4Error):
        slz.lz4FrameDecompress(bytes(f), eng)


def _tiny_block_frame(n_blocks, bd=0x70):
    """A frame that NAMES 4 MiB per block and holds one literal per block: [size 2][token 0x10][byte]."""
    import streamly_lz4_amd as slz
    desc = bytes([0x60, bd])                                    # version 01, independent blocks; BD: 4 MiB
    hc = (slz.xxh32(desc) >> 8) & 0xFF
    body = b"".join(struct.pack("<I", 2) + bytes([0x10, 65 + (k % 26)]) for k in range(n_blocks))
    return struct.pack("<I", 0x184D2204) + desc + bytes([hc]) + body + struct.pack("<I", 0)


def test_crafted_frame_cannot_reserve_more_than_it_decodes(eng):
    """Round-2 advisor finding: the output used to be sized as blocks x maximum block size before anything was
    decoded -- 20 000 two-byte blocks with BD = 4 MiB asked for 80 GB on the host and on the device.  Blocks are
    now decoded in groups under a byte budget, each block capped by what its bytes can expand to."""
    import resource
    import streamly_lz4_amd as slz
    n = 20000
    frame = _tiny_block_frame(n)
    before = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    out = slz.lz4FrameDecompress(frame, eng)
    after = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    assert out == bytes(65 + (k % 26) for k in range(n))
    assert (after - before) < (1 << 20), "peak RSS grew by more than 1 GiB (KiB units): %d" % (after - before)


@needs_lz4f
def test_linked_frame_longer_than_one_group(eng):
    """A linked frame whose blocks do not fit one group (4 MiB blocks, 256 MiB budget): the window crosses the seam
    as the next call's dictionary."""
    import streamly_lz4_amd as slz
    data = _text(70 * (4 << 20) // 16, 21) * 16                 # 280 MiB of text with long-range repeats
    frame = lz4f.compress_frame(L, data, block_id=7, linked=True, content_checksum=True)
    assert slz.lz4FrameDecompress(frame, eng) == data
