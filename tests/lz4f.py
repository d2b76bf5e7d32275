"""ctypes binding of the system liblz4's LZ4F_* frame API (lz4frame.h, v1.9.3 in this image) -- the independent
implementation the frame-format tests interoperate with.  Test support only; absent library => tests skip."""
import ctypes as C
import ctypes.util


class FrameInfo(C.Structure):
    _fields_ = [("blockSizeID", C.c_int), ("blockMode", C.c_int), ("contentChecksumFlag", C.c_int),
                ("frameType", C.c_int), ("contentSize", C.c_ulonglong), ("dictID", C.c_uint),
                ("blockChecksumFlag", C.c_int)]


class Preferences(C.Structure):
    _fields_ = [("frameInfo", FrameInfo), ("compressionLevel", C.c_int), ("autoFlush", C.c_uint),
                ("favorDecSpeed", C.c_uint), ("reserved", C.c_uint * 3)]


def load():
    for name in ("liblz4.so.1", ctypes.util.find_library("lz4")):
        if not name:
            continue
        try:
            L = C.CDLL(name)
        except OSError:
            continue
        if not hasattr(L, "LZ4F_compressFrame"):
            continue
        vp, sz = C.c_void_p, C.c_size_t
        L.LZ4F_isError.restype = C.c_uint
        L.LZ4F_isError.argtypes = [sz]
        L.LZ4F_getErrorName.restype = C.c_char_p
        L.LZ4F_getErrorName.argtypes = [sz]
        L.LZ4F_compressFrameBound.restype = sz
        L.LZ4F_compressFrameBound.argtypes = [sz, C.POINTER(Preferences)]
        L.LZ4F_compressFrame.restype = sz
        L.LZ4F_compressFrame.argtypes = [vp, sz, vp, sz, C.POINTER(Preferences)]
        L.LZ4F_createCompressionContext.restype = sz
        L.LZ4F_createCompressionContext.argtypes = [C.POINTER(vp), C.c_uint]
        L.LZ4F_freeCompressionContext.restype = sz
        L.LZ4F_freeCompressionContext.argtypes = [vp]
        L.LZ4F_compressBegin.restype = sz
        L.LZ4F_compressBegin.argtypes = [vp, vp, sz, C.POINTER(Preferences)]
        L.LZ4F_compressBound.restype = sz
        L.LZ4F_compressBound.argtypes = [sz, C.POINTER(Preferences)]
        L.LZ4F_compressUpdate.restype = sz
        L.LZ4F_compressUpdate.argtypes = [vp, vp, sz, vp, sz, vp]
        L.LZ4F_flush.restype = sz
        L.LZ4F_flush.argtypes = [vp, vp, sz, vp]
        L.LZ4F_compressEnd.restype = sz
        L.LZ4F_compressEnd.argtypes = [vp, vp, sz, vp]
        L.LZ4F_createDecompressionContext.restype = sz
        L.LZ4F_createDecompressionContext.argtypes = [C.POINTER(vp), C.c_uint]
        L.LZ4F_freeDecompressionContext.restype = sz
        L.LZ4F_freeDecompressionContext.argtypes = [vp]
        L.LZ4F_decompress.restype = sz
        L.LZ4F_decompress.argtypes = [vp, vp, C.POINTER(sz), vp, C.POINTER(sz), vp]
        return L
    return None


class LZ4FError(RuntimeError):
    pass


def _check(L, code, what):
    if L.LZ4F_isError(code):
        raise LZ4FError("%s: %s" % (what, L.LZ4F_getErrorName(code).decode()))
    return code


def prefs(block_id=4, linked=False, content_checksum=False, block_checksum=False, content_size=0, level=0):
    p = Preferences()
    p.frameInfo.blockSizeID = block_id            # 4 = 64 KiB ... 7 = 4 MiB
    p.frameInfo.blockMode = 0 if linked else 1
    p.frameInfo.contentChecksumFlag = int(content_checksum)
    p.frameInfo.blockChecksumFlag = int(block_checksum)
    p.frameInfo.contentSize = content_size
    p.compressionLevel = level
    return p


def compress_frame(L, data, **kw):
    """LZ4F_compressFrame (one shot)."""
    p = prefs(**kw)
    cap = L.LZ4F_compressFrameBound(len(data), C.byref(p))
    dst = C.create_string_buffer(cap)
    n = _check(L, L.LZ4F_compressFrame(dst, cap, data, len(data), C.byref(p)), "LZ4F_compressFrame")
    return dst.raw[:n]


def compress_pieces(L, pieces, **kw):
    """Streaming writer that flushes behind every piece: short blocks in mid-frame."""
    p = prefs(**kw)
    ctx = C.c_void_p()
    _check(L, L.LZ4F_createCompressionContext(C.byref(ctx), 100), "createCompressionContext")
    try:
        out = bytearray()
        cap = max(L.LZ4F_compressBound(max((len(x) for x in pieces), default=0), C.byref(p)), 64) + 64
        dst = C.create_string_buffer(cap)

        def take(n):                               # (n is computed before dst is read)
            out.extend(dst.raw[:n])

        take(_check(L, L.LZ4F_compressBegin(ctx, dst, cap, C.byref(p)), "compressBegin"))
        for piece in pieces:
            take(_check(L, L.LZ4F_compressUpdate(ctx, dst, cap, piece, len(piece), None), "compressUpdate"))
            take(_check(L, L.LZ4F_flush(ctx, dst, cap, None), "flush"))
        take(_check(L, L.LZ4F_compressEnd(ctx, dst, cap, None), "compressEnd"))
        return bytes(out)
    finally:
        L.LZ4F_freeCompressionContext(ctx)


def decompress(L, frame, max_out):
    """LZ4F_decompress over a whole buffer (one or more frames)."""
    ctx = C.c_void_p()
    _check(L, L.LZ4F_createDecompressionContext(C.byref(ctx), 100), "createDecompressionContext")
    try:
        src = C.create_string_buffer(bytes(frame), len(frame))
        dst = C.create_string_buffer(max(max_out, 1))
        spos, dpos = 0, 0
        while spos < len(frame):
            dsz = C.c_size_t(max_out - dpos)
            ssz = C.c_size_t(len(frame) - spos)
            hint = _check(L, L.LZ4F_decompress(ctx, C.byref(dst, dpos), C.byref(dsz), C.byref(src, spos), C.byref(ssz), None),
                          "LZ4F_decompress")
            spos += ssz.value
            dpos += dsz.value
            if ssz.value == 0 and dsz.value == 0:
                if hint != 0:
                    raise LZ4FError("LZ4F_decompress: no progress (output buffer of %d too small or truncated frame)" % max_out)
                break
        return dst.raw[:dpos]
    finally:
        L.LZ4F_freeDecompressionContext(ctx)
