"""Pure-Python restatement of the reference's framing layer (host side of the hot path).

TEST INFRASTRUCTURE ONLY (see oracle/README.md).  Used to check the C++ mirror in
streamly-lz4_amd/csrc/host_stream.cpp.  Line numbers: src/Streamly/Internal/LZ4.hs.
"""
import struct


class RefError(Exception):
    """Stands for the reference's `error` / `Parser.die`."""


def meta_size(has_size):  # :177-181
    return 8 if has_size else 4


def frame_block(comp, uncomp_len, has_size=True):
    """Header layout written by compressChunk (:261-262): LE int32 compLen @0, LE int32 uncompLen @4."""
    if has_size:
        return struct.pack("<ii", len(comp), uncomp_len) + comp
    return struct.pack("<i", len(comp)) + comp


def split_frames(framed, has_size=True):
    """Walk a dense framed stream -> list of (compLen, uncompLen or None, payload)."""
    out, pos, meta = [], 0, meta_size(has_size)
    while pos < len(framed):
        c = struct.unpack_from("<i", framed, pos)[0]
        u = struct.unpack_from("<i", framed, pos + 4)[0] if has_size else None
        out.append((c, u, framed[pos + meta:pos + meta + c]))
        pos += meta + c
    return out


def resize_chunks(arrays, has_size=True, has_end_mark=False):
    """resizeChunksD (:432-523) as a generator-free state machine over a list of bytes."""
    meta = meta_size(has_size)
    footer = 4 if has_end_mark else 0  # :404-408
    it = iter(arrays)
    out = []
    state, buf = "init", b""
    while True:
        if state == "init":  # RInit :488-496
            try:
                buf = bytes(next(it))
            except StopIteration:
                if has_end_mark:
                    raise RefError("resizeChunksD: No end mark found")
                return out
            state = "process"
        elif state == "process":  # :459-484
            ln = len(buf)
            if ln < 4:
                state = "accumulate"
            elif has_end_mark and struct.unpack_from("<i", buf, 0)[0] == 0:  # :451-456
                state = "footer"
            elif ln <= meta:
                state = "accumulate"
            else:
                required = struct.unpack_from("<i", buf, 0)[0] + meta
                if ln == required:
                    out.append(buf)
                    state = "init"
                elif ln < required:
                    state = "accumulate"
                else:
                    out.append(buf[:required])
                    buf = buf[required:]
        elif state == "accumulate":  # :498-505
            try:
                buf = buf + bytes(next(it))
            except StopIteration:
                raise RefError("resizeChunksD: Incomplete block")
            state = "process"
        elif state == "footer":  # :506-522
            if len(buf) < footer:
                try:
                    buf = buf + bytes(next(it))
                except StopIteration:
                    raise RefError("resizeChunksD: Incomplete footer")
            else:
                return out  # validateFooter is always True (:410-411)


FRAME_MAGIC = 407708164  # 0x184D2204, :608


def simple_frame_parser(data):
    """simpleFrameParserD (:590-651) over a bytes object.  Returns (block_max, rest)."""
    if len(data) < 7:
        raise RefError("unexpected end of input")
    magic = struct.unpack_from("<I", data, 0)[0]
    if magic != FRAME_MAGIC:
        raise RefError("The parsed magic %d does not match %d" % (magic, FRAME_MAGIC))
    flg = data[4]
    if not ((flg & 0x80) == 0 and (flg & 0x40) != 0):
        raise RefError("Version is not 01")
    for bit, msg in ((0x20, "Block independence"), (0x10, "Block checksum"), (0x08, "Content size"),
                     (0x04, "Content checksum"), (0x01, "Dict")):
        if flg & bit:
            raise RefError(msg + " is not yet supported")
    bd = data[5] >> 4
    sizes = {4: 64 << 10, 5: 256 << 10, 6: 1 << 20, 7: 4 << 20}
    if bd not in sizes:
        raise RefError("parseBD: Unknown block max size")
    return sizes[bd], data[7:]  # data[6] = header checksum, unchecked (:605)
