"""streamly_lz4_amd -- Python binding of the MI355X LZ4 block engine.

Thin ctypes layer over the C ABI in ``include/mi355lz4.h`` (device-resident and
host-buffer batched calls) and over the C++ mirror of the reference's stream
combinators (``include/streamly_lz4.hpp``: compressChunks / decompressChunks /
resizeChunks / decompressChunksWith; reference src/Streamly/LZ4.hs:94-122,
src/Streamly/Internal/LZ4.hs:338-651).

There is no CPU codec in this package: if ``libmi355lz4.so`` is missing, or no
gfx950 device is visible, every codec call raises.  PyTorch is used only by
callers that want device memory / streams / torch.distributed; this module
itself needs only ctypes and numpy.
"""
import ctypes as C
import sys
import os

import numpy as np

__all__ = [
    "lib", "lib_path", "Engine", "LZ4Error", "BlockSize", "BlockConfig", "FrameConfig",
    "defaultBlockConfig", "defaultFrameConfig", "setBlockMaxSize", "setFrameEndMark",
    "compressChunks", "decompressChunks", "decompressChunksRaw", "resizeChunks",
    "decompressChunksWith", "simpleFrameParser", "compress_bound", "slot_stride", "device_count",
    "xxh32", "lz4FrameCompress", "lz4FrameDecompress",
]

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
# MI355LZ4_LIB lets a developer A/B another build of the same library (scripts/ab_variants.sh)
lib_path = os.environ.get("MI355LZ4_LIB") or os.path.join(os.path.dirname(_PKG_DIR), "lib", "libmi355lz4.so")

_u8p = C.POINTER(C.c_uint8)
_u64p = C.POINTER(C.c_uint64)
_i32p = C.POINTER(C.c_int32)


class LZ4Error(RuntimeError):
    """Raised where the reference calls `error` / `Parser.die`, with the same message."""


def _preload_torch_hip():
    """PyTorch wheels bundle their own HIP/HSA runtime (same SONAME as /opt/rocm's).  Two HSA
    runtimes in one process do not coexist, so when torch is installed make its copy the one that
    is loaded, whichever of torch / this package is imported first."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        p = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(p):
            C.CDLL(p, mode=os.RTLD_GLOBAL)
    except Exception:
        pass


def _load():
    _preload_torch_hip()
    if not os.path.exists(lib_path):
        raise ImportError(
            "streamly_lz4_amd: %s is missing -- build it with `make lib` (hipcc, gfx950). "
            "There is no CPU fallback." % lib_path
        )
    L = C.CDLL(lib_path, mode=os.RTLD_LOCAL)

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    vp = C.c_void_p
    sig("mi355lz4_version", C.c_int)
    sig("mi355lz4_last_error", C.c_char_p)
    sig("mi355lz4_device_count", C.c_int)
    sig("mi355lz4_create", C.c_int, C.POINTER(vp), C.c_int)
    sig("mi355lz4_destroy", None, vp)
    sig("mi355lz4_set_stream", C.c_int, vp, vp)
    sig("mi355lz4_get_stream", vp, vp)
    sig("mi355lz4_synchronize", C.c_int, vp)
    sig("mi355lz4_set_decoder", C.c_int, vp, C.c_int)
    sig("mi355lz4_set_segments", C.c_int, vp, C.c_int)
    sig("mi355lz4_set_linked_async", C.c_int, vp, C.c_int)
    sig("mi355lz4_set_linked_compress", C.c_int, vp, C.c_int)
    sig("mi355lz4_compress_bound", C.c_int, C.c_int)
    sig("mi355lz4_slot_stride", C.c_size_t, C.c_int, C.c_int)
    sig("mi355lz4_compress_batch_device", C.c_int, vp, vp, vp, vp, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int,
        vp, C.c_size_t, vp)
    sig("mi355lz4_compact_device", C.c_int, vp, vp, C.c_size_t, vp, C.c_int, vp, C.c_size_t, vp)
    sig("mi355lz4_decompress_batch_device", C.c_int, vp, vp, C.c_uint64, vp, C.c_int, C.c_int, C.c_int, C.c_int,
        vp, vp, vp, vp)
    sig("mi355lz4_decompress_streams_device", C.c_int, vp, vp, C.c_uint64, vp, C.c_int, C.c_int, C.c_int, vp,
        C.c_int, vp, vp, vp, vp)
    sig("mi355lz4_decompress_linked_begin", C.c_int, vp, vp, C.c_uint64, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp,
        C.c_int)
    sig("mi355lz4_decompress_linked_end", C.c_int, vp)
    sig("mi355lz4_decompress_linked_end_last", C.c_int, vp)
    sig("mi355lz4_index_device", C.c_int, vp, vp, C.c_uint64, vp, C.c_int, C.c_int, C.c_int, vp)
    sig("mi355lz4_compress_batch", C.c_int, vp, C.POINTER(_u8p), _i32p, C.c_int, C.c_int, C.c_int, _u8p,
        C.c_size_t, C.POINTER(C.c_size_t), _i32p, _i32p)
    sig("mi355lz4_index_host", C.c_int, _u8p, C.c_size_t, C.c_int, C.c_int, _u64p, _i32p, C.c_int,
        C.POINTER(C.c_int))
    sig("mi355lz4_decompress_batch", C.c_int, vp, _u8p, C.c_size_t, C.c_int, C.c_int, C.c_int, _u8p, C.c_int,
        _u8p, C.c_size_t, C.POINTER(C.c_size_t), _i32p, C.c_int, C.POINTER(C.c_int))
    sig("mi355lz4_decompress_streams", C.c_int, vp, _u8p, C.c_size_t, C.c_int, C.c_int, _i32p, C.c_int,
        _u8p, C.c_size_t, C.POINTER(C.c_size_t), _i32p, C.c_int, C.POINTER(C.c_int))
    sig("mi355lz4_generate_device", C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_uint64, C.c_uint64,
        C.c_uint32, C.c_uint32)
    sig("mi355lz4_interleave_device", C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp)
    sig("mi355lz4_event_create", C.c_int, C.POINTER(vp))
    sig("mi355lz4_event_destroy", C.c_int, vp)
    sig("mi355lz4_event_record", C.c_int, vp, vp)
    sig("mi355lz4_event_elapsed_ms", C.c_int, vp, vp, C.POINTER(C.c_float))
    # legacy face (include/lz4.h)
    sig("LZ4_createStream", vp)
    sig("LZ4_freeStream", C.c_int, vp)
    sig("LZ4_createStreamDecode", vp)
    sig("LZ4_freeStreamDecode", C.c_int, vp)
    sig("LZ4_compressBound", C.c_int, C.c_int)
    sig("LZ4_compress_fast_continue", C.c_int, vp, _u8p, _u8p, C.c_int, C.c_int, C.c_int)
    sig("LZ4_decompress_safe_continue", C.c_int, vp, _u8p, _u8p, C.c_int, C.c_int)
    # stream-combinator mirror
    sig("slz4_last_error", C.c_char_p)
    sig("slz4_engine_create", C.c_int, C.POINTER(vp), C.c_int, C.c_size_t)
    sig("slz4_engine_destroy", None, vp)
    sig("slz4_engine_set_batch", None, vp, C.c_size_t)
    sig("slz4_engine_set_linked_compress", None, vp, C.c_int)
    sig("slz4_engine_ctx", vp, vp)
    sig("slz4_arrays_count", C.c_size_t, vp)
    sig("slz4_arrays_flat", C.c_int, vp, C.POINTER(_u8p), C.POINTER(C.POINTER(C.c_size_t)))
    sig("slz4_arrays_len", C.c_size_t, vp, C.c_size_t)
    sig("slz4_arrays_data", _u8p, vp, C.c_size_t)
    sig("slz4_arrays_free", None, vp)
    sig("slz4_trim", None)
    sig("slz4_compress_chunks", C.c_int, vp, C.c_int, C.c_int, _u8p, _u64p, C.c_size_t, C.POINTER(vp))
    sig("slz4_resize_chunks", C.c_int, C.c_int, C.c_int, _u8p, _u64p, C.c_size_t, C.POINTER(vp))
    sig("slz4_decompress_chunks_raw", C.c_int, vp, C.c_int, _u8p, _u64p, C.c_size_t, C.POINTER(vp))
    sig("slz4_decompress_chunks", C.c_int, vp, C.c_int, C.c_int, _u8p, _u64p, C.c_size_t, C.POINTER(vp))
    sig("slz4_decompress_chunks_with", C.c_int, vp, _u8p, _u64p, C.c_size_t, C.POINTER(vp))
    sig("slz4_simple_frame_parser", C.c_int, _u8p, _u64p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(vp))
    # standard LZ4 frames
    sig("slz4_xxh32", C.c_uint32, _u8p, C.c_size_t, C.c_uint32)
    sig("slz4_lz4frame_compress", C.c_int, vp, C.c_int, C.c_int, C.c_int, _u8p, C.c_size_t, C.POINTER(vp))
    sig("slz4_lz4frame_decompress", C.c_int, vp, _u8p, C.c_size_t, C.POINTER(vp))
    return L


lib = _load()

# every symbol include/mi355lz4.h and include/lz4.h declare (checked by tests without a GPU)
DECLARED_SYMBOLS = [
    "mi355lz4_version", "mi355lz4_last_error", "mi355lz4_device_count", "mi355lz4_create", "mi355lz4_destroy",
    "mi355lz4_set_stream", "mi355lz4_get_stream", "mi355lz4_synchronize", "mi355lz4_set_decoder", "mi355lz4_set_segments", "mi355lz4_set_linked_async", "mi355lz4_set_linked_compress",
    "mi355lz4_compress_bound", "mi355lz4_slot_stride", "mi355lz4_compress_batch_device", "mi355lz4_compact_device",
    "mi355lz4_decompress_batch_device", "mi355lz4_decompress_streams_device", "mi355lz4_decompress_linked_begin",
    "mi355lz4_decompress_linked_end", "mi355lz4_decompress_linked_end_last", "mi355lz4_index_device", "mi355lz4_compress_batch", "mi355lz4_index_host",
    "mi355lz4_decompress_batch", "mi355lz4_decompress_streams", "mi355lz4_generate_device", "mi355lz4_interleave_device", "mi355lz4_event_create",
    "mi355lz4_event_destroy", "mi355lz4_event_record", "mi355lz4_event_elapsed_ms",
    "mi355lz4_create_multi", "mi355lz4_destroy_multi", "mi355lz4_multi_device_count", "mi355lz4_multi_engine", "mi355lz4_multi_last_error",
    "mi355lz4_multi_compress_batch", "mi355lz4_multi_decompress_batch",
    "LZ4_createStream", "LZ4_freeStream", "LZ4_createStreamDecode", "LZ4_freeStreamDecode", "LZ4_compressBound",
    "LZ4_compress_fast_continue", "LZ4_decompress_safe_continue",
]


def _err():
    return (lib.mi355lz4_last_error() or b"").decode("utf-8", "replace")


def _check(rc, what=""):
    if rc != 0:
        raise LZ4Error("%s failed (%d): %s" % (what or "mi355lz4 call", rc, _err()))


def compress_bound(n):
    return lib.mi355lz4_compress_bound(int(n))


def slot_stride(block_len, header_kind=8):
    return lib.mi355lz4_slot_stride(int(block_len), int(header_kind))


def device_count():
    return lib.mi355lz4_device_count()


# ---------------------------------------------------------------------------
# Config mirror (reference src/Streamly/Internal/LZ4/Config.hs)
# ---------------------------------------------------------------------------
class BlockSize:
    BlockHasSize = 0
    BlockMax64KB = 1
    BlockMax256KB = 2
    BlockMax1MB = 3
    BlockMax4MB = 4


class BlockConfig:
    def __init__(self, blockSize=BlockSize.BlockHasSize):
        self.blockSize = blockSize

    @property
    def metaSize(self):  # Internal/LZ4.hs:177-181
        return 8 if self.blockSize == BlockSize.BlockHasSize else 4

    @property
    def fixedUncomp(self):  # Internal/LZ4.hs:189-198
        return {0: 0, 1: 64 << 10, 2: 256 << 10, 3: 1 << 20, 4: 4 << 20}[self.blockSize]


class FrameConfig:
    def __init__(self, hasEndMark=False):
        self.hasEndMark = hasEndMark


defaultBlockConfig = BlockConfig()
defaultFrameConfig = FrameConfig()


def setBlockMaxSize(bs, cfg):
    return BlockConfig(bs)


def setFrameEndMark(v, cfg):
    return FrameConfig(bool(v))


# ---------------------------------------------------------------------------
# Engine
# ---------------------------------------------------------------------------
def _dptr(x):
    """Device pointer of a torch tensor / int / None."""
    if x is None:
        return None
    if isinstance(x, int):
        return C.c_void_p(x)
    return C.c_void_p(x.data_ptr())


class Event:
    def __init__(self):
        self.h = C.c_void_p()
        _check(lib.mi355lz4_event_create(C.byref(self.h)), "event_create")

    def __del__(self):
        try:
            if self.h:
                lib.mi355lz4_event_destroy(self.h)
        except Exception:
            pass


class MultiEngine:
    """Several GPUs behind one handle, one process (include/mi355lz4.h, "several GPUs behind one handle"): the host-buffer
    calls of Engine with the batch cut into one contiguous block range per device.  devices: list of HIP device ordinals
    (the same one may be named more than once: two engines on one GPU)."""

    def __init__(self, devices):
        self._h = C.c_void_p()
        arr = (C.c_int * len(devices))(*[int(d) for d in devices])
        rc = lib.mi355lz4_create_multi(C.byref(self._h), arr, len(devices))
        if rc != 0:
            lib.mi355lz4_multi_last_error.restype = C.c_char_p
            raise LZ4Error("create_multi failed (%d): %s" % (rc, (lib.mi355lz4_multi_last_error() or b"").decode()))
        self.n = len(devices)

    def close(self):
        if self._h:
            lib.mi355lz4_destroy_multi(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _raise(self, rc, what, allow_block_error):
        if rc != 0 and not (allow_block_error and rc == -5):
            lib.mi355lz4_multi_last_error.restype = C.c_char_p
            raise LZ4Error("%s failed (%d): %s" % (what, rc, (lib.mi355lz4_multi_last_error() or b"").decode()))

    def set_decoder(self, variant):
        lib.mi355lz4_multi_engine.restype = C.c_void_p
        for i in range(self.n):
            _check(lib.mi355lz4_set_decoder(C.c_void_p(lib.mi355lz4_multi_engine(self._h, i)), int(variant)), "set_decoder")

    def compress_batch(self, blocks, accel=1, header_kind=8):
        """blocks: list of bytes-like.  Returns (framed bytes, [framed length per block])."""
        n = len(blocks)
        arrs = [np.frombuffer(bytes(b), dtype=np.uint8) if not isinstance(b, np.ndarray) else b for b in blocks]
        ptrs = (_u8p * max(n, 1))(*[a.ctypes.data_as(_u8p) for a in arrs])
        lens = np.array([a.size for a in arrs], dtype=np.int32)
        cap = int(sum(compress_bound(int(x)) + header_kind for x in lens)) + 16
        out = np.empty(cap, dtype=np.uint8)
        out_len = C.c_size_t()
        flen = np.zeros(max(n, 1), dtype=np.int32)
        status = np.zeros(max(n, 1), dtype=np.int32)
        rc = lib.mi355lz4_multi_compress_batch(self._h, ptrs, lens.ctypes.data_as(_i32p), n, int(accel), int(header_kind),
                                               out.ctypes.data_as(_u8p), C.c_size_t(cap), C.byref(out_len),
                                               flen.ctypes.data_as(_i32p), status.ctypes.data_as(_i32p))
        self._raise(rc, "multi_compress_batch", False)
        return out[: out_len.value].tobytes(), flen[:n].tolist()

    def decompress_batch(self, framed, header_kind=8, fixed_uncomp=0, raise_on_block_error=True):
        """Returns (decoded bytes, [decoded length or negative code per block])."""
        src = np.frombuffer(bytes(framed), dtype=np.uint8)
        max_blocks = src.size // (header_kind + 1) + 1
        boff = np.zeros(max_blocks + 1, dtype=np.uint64)
        ulen = np.zeros(max_blocks + 1, dtype=np.int32)
        nb = C.c_int()
        _check(lib.mi355lz4_index_host(src.ctypes.data_as(_u8p), src.size, header_kind, fixed_uncomp,
                                       boff.ctypes.data_as(_u64p), ulen.ctypes.data_as(_i32p), max_blocks, C.byref(nb)), "index_host")
        cap = int(ulen[: nb.value].astype(np.int64).clip(min=0).sum()) + 16
        out = np.empty(cap, dtype=np.uint8)
        out_len = C.c_size_t()
        blen = np.zeros(max(nb.value, 1), dtype=np.int32)
        got = C.c_int()
        rc = lib.mi355lz4_multi_decompress_batch(self._h, src.ctypes.data_as(_u8p), C.c_size_t(src.size), header_kind, fixed_uncomp,
                                                 out.ctypes.data_as(_u8p), C.c_size_t(cap), C.byref(out_len),
                                                 blen.ctypes.data_as(_i32p), max(nb.value, 1), C.byref(got))
        self._raise(rc, "multi_decompress_batch", not raise_on_block_error)
        return out[: out_len.value].tobytes(), blen[: got.value].tolist()


class Engine:
    """One GPU engine (HIP device + stream).  Wraps mi355lz4_ctx and the C++ stream-combinator engine."""

    def __init__(self, device=0, batch_blocks=4096):
        self._h = C.c_void_p()
        if lib.slz4_engine_create(C.byref(self._h), int(device), int(batch_blocks)) != 0:
            raise LZ4Error((lib.slz4_last_error() or b"").decode())
        self.ctx = C.c_void_p(lib.slz4_engine_ctx(self._h))
        self.device = device
        # The engine creates a non-blocking stream of its own.  The tensors this binding is handed are produced
        # and consumed by torch on torch's CURRENT stream, so every device-API call launches there: the stream is
        # re-read on each call (it changes under `with torch.cuda.stream(s):`), and a `torch.zeros(...)` followed
        # by a decode into that tensor is ordered without an explicit synchronisation.  use_stream() pins the
        # engine to one stream instead.
        self._pinned = False
        self._cur_stream = None
        self._follow_torch()

    def _follow_torch(self):
        """Launch on torch's current stream for this device (torch may be imported after the engine)."""
        if self._pinned:
            return
        torch = sys.modules.get("torch")
        if torch is not None and torch.cuda.is_available():
            s = torch.cuda.current_stream(self.device).cuda_stream
            if s != self._cur_stream:
                _check(lib.mi355lz4_set_stream(self.ctx, C.c_void_p(s)), "set_stream")
                self._cur_stream = s

    def close(self):
        if self._h:
            lib.slz4_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_batch_blocks(self, n):
        lib.slz4_engine_set_batch(self._h, int(n))

    @staticmethod
    def has_experiments():
        """True when the loaded library is the experiment build (make lib-exp): decoder variant 3 exists."""
        try:
            f = lib.mi355lz4_debug_has_experiments
        except AttributeError:
            return False
        f.restype = C.c_int
        return bool(f())

    def set_decoder(self, variant):
        _check(lib.mi355lz4_set_decoder(self.ctx, int(variant)), "set_decoder")

    def set_linked_compress(self, on):
        """Compress calls write ONE linked stream (previous block = dictionary), like the reference's compressor."""
        lib.slz4_engine_set_linked_compress(self._h, int(bool(on)))

    def set_linked_async(self, max_decoded_block_size):
        """Linked device decodes enqueue only, no host wait (include/mi355lz4.h); 0 = default (wait)."""
        _check(lib.mi355lz4_set_linked_async(self.ctx, int(max_decoded_block_size)), "set_linked_async")

    def set_segments(self, segs):
        """Small-batch compression: segments per block (-1 automatic, 0 off, 2..64 forced); include/mi355lz4.h."""
        _check(lib.mi355lz4_set_segments(self.ctx, int(segs)), "set_segments")

    def use_stream(self, hip_stream):
        """Pin the engine to one caller-owned hipStream_t (stops following torch's current stream)."""
        _check(lib.mi355lz4_set_stream(self.ctx, C.c_void_p(hip_stream)), "set_stream")
        self._pinned = True
        self._cur_stream = hip_stream

    def synchronize(self):
        _check(lib.mi355lz4_synchronize(self.ctx), "synchronize")

    # ---- timing on the engine's own stream ---------------------------------
    def record(self, ev):
        self._follow_torch()
        _check(lib.mi355lz4_event_record(self.ctx, ev.h), "event_record")

    @staticmethod
    def elapsed_ms(start, stop):
        ms = C.c_float()
        _check(lib.mi355lz4_event_elapsed_ms(start.h, stop.h, C.byref(ms)), "event_elapsed")
        return ms.value

    # ---- device-resident batched API (torch uint8/int32/int64 CUDA tensors) ----
    def generate(self, kind, dst, block_len, n_blocks, first_block=0, block_step=1, lit_max=16, off_max=2048):
        self._follow_torch()
        k = {"random": 0, "lzsynth": 1, "text": 2}[kind]
        _check(lib.mi355lz4_generate_device(self.ctx, k, _dptr(dst), int(block_len), int(n_blocks), int(first_block),
                                            int(block_step), int(lit_max), int(off_max)), "generate_device")

    def compress_batch_device(self, src, n_blocks, max_block_len, slots, slot_stride_, framed_len, accel=1,
                              header_kind=8, src_off=None, src_len=None, block_stride=None):
        self._follow_torch()
        _check(lib.mi355lz4_compress_batch_device(
            self.ctx, _dptr(src), _dptr(src_off), _dptr(src_len),
            int(max_block_len if block_stride is None else block_stride), int(max_block_len), int(n_blocks),
            int(accel), int(header_kind), _dptr(slots), int(slot_stride_), _dptr(framed_len)), "compress_batch_device")

    def compact_device(self, slots, slot_stride_, framed_len, n_blocks, dense, dense_cap, dense_off):
        self._follow_torch()
        _check(lib.mi355lz4_compact_device(self.ctx, _dptr(slots), int(slot_stride_), _dptr(framed_len), int(n_blocks),
                                           _dptr(dense), int(dense_cap), _dptr(dense_off)), "compact_device")

    def decompress_batch_device(self, framed, framed_len, block_off, n_blocks, out, out_off, result, header_kind=8,
                                fixed_uncomp=0, linked=False, out_cap=None):
        self._follow_torch()
        _check(lib.mi355lz4_decompress_batch_device(
            self.ctx, _dptr(framed), int(framed_len), _dptr(block_off), int(n_blocks), int(header_kind),
            int(fixed_uncomp), int(bool(linked)), _dptr(out), _dptr(out_off), _dptr(out_cap), _dptr(result)),
            "decompress_batch_device")

    def decompress_streams_device(self, framed, framed_len, block_off, n_blocks, stream_first, n_streams, out,
                                  out_off, result, header_kind=8, fixed_uncomp=0, out_cap=None):
        self._follow_torch()
        """Linked streams: stream s = blocks [stream_first[s], stream_first[s+1]) (int32 device tensor)."""
        _check(lib.mi355lz4_decompress_streams_device(
            self.ctx, _dptr(framed), int(framed_len), _dptr(block_off), int(n_blocks), int(header_kind),
            int(fixed_uncomp), _dptr(stream_first), int(n_streams), _dptr(out), _dptr(out_off), _dptr(out_cap),
            _dptr(result)), "decompress_streams_device")

    def decompress_linked_begin(self, framed, framed_len, block_off, n_blocks, out, out_off, result, look_back,
                                header_kind=8, fixed_uncomp=0):
        """First half of a linked decode of a contiguous RANGE of one stream (include/mi355lz4.h): everything that does
        not read the output of the block in front of the range.  With look_back = 1, out_off (int64, n_blocks + 2) and
        result (int32, n_blocks + 1) carry one leading entry for that block: where its output will be and its size."""
        self._follow_torch()
        lb = 1 if look_back else 0
        _check(lib.mi355lz4_decompress_linked_begin(
            self.ctx, _dptr(framed), int(framed_len), _dptr(block_off), int(n_blocks), int(header_kind), int(fixed_uncomp),
            _dptr(out), C.c_void_p(out_off.data_ptr() + 8 * lb), None, C.c_void_p(result.data_ptr() + 4 * lb), lb),
            "decompress_linked_begin")

    def decompress_linked_end_last(self):
        """Between begin and end: make the range's LAST block final ahead of the rest.  True: its bytes and result are
        final (hand them on, call decompress_linked_end afterwards); False: not available this way, call
        decompress_linked_end first."""
        self._follow_torch()
        r = lib.mi355lz4_decompress_linked_end_last(self.ctx)
        if r < 0:
            _check(r, "decompress_linked_end_last")
        return r == 1

    def decompress_linked_end(self):
        """Second half: fetch from the roots (the first of which lie in the look-back block's output), results."""
        self._follow_torch()
        _check(lib.mi355lz4_decompress_linked_end(self.ctx), "decompress_linked_end")

    def index_device(self, framed, framed_len, block_off, n_blocks, out_off, header_kind=8, fixed_uncomp=0):
        self._follow_torch()
        _check(lib.mi355lz4_index_device(self.ctx, _dptr(framed), int(framed_len), _dptr(block_off), int(n_blocks),
                                         int(header_kind), int(fixed_uncomp), _dptr(out_off)), "index_device")

    def interleave_device(self, local, local_off, n_local, rank, n_ranks, global_buf, global_off):
        self._follow_torch()
        _check(lib.mi355lz4_interleave_device(self.ctx, _dptr(local), _dptr(local_off), int(n_local), int(rank),
                                              int(n_ranks), _dptr(global_buf), _dptr(global_off)), "interleave_device")

    # ---- host-buffer batched API -------------------------------------------
    def compress_batch(self, blocks, accel=1, header_kind=8):
        """blocks: list of bytes-like.  Returns (framed bytes, [framed length per block])."""
        n = len(blocks)
        arrs = [np.frombuffer(bytes(b), dtype=np.uint8) if not isinstance(b, np.ndarray) else b for b in blocks]
        ptrs = (_u8p * max(n, 1))(*[a.ctypes.data_as(_u8p) for a in arrs])
        lens = np.array([a.size for a in arrs], dtype=np.int32)
        cap = int(sum(compress_bound(int(x)) + header_kind for x in lens)) + 16
        out = np.empty(cap, dtype=np.uint8)
        out_len = C.c_size_t()
        flen = np.zeros(max(n, 1), dtype=np.int32)
        status = np.zeros(max(n, 1), dtype=np.int32)
        _check(lib.mi355lz4_compress_batch(self.ctx, ptrs, lens.ctypes.data_as(_i32p), n, int(accel), int(header_kind),
                                           out.ctypes.data_as(_u8p), cap, C.byref(out_len),
                                           flen.ctypes.data_as(_i32p), status.ctypes.data_as(_i32p)), "compress_batch")
        return out[: out_len.value].tobytes(), flen[:n].tolist()

    def decompress_batch(self, framed, header_kind=8, fixed_uncomp=0, linked=False, dict_bytes=None, max_blocks=None,
                         raise_on_block_error=True):
        """Returns (decoded bytes, [decoded length or negative code per block])."""
        src = np.frombuffer(bytes(framed), dtype=np.uint8)
        if max_blocks is None:
            max_blocks = src.size // (header_kind + 1) + 1
        boff = np.zeros(max_blocks + 1, dtype=np.uint64)
        ulen = np.zeros(max_blocks + 1, dtype=np.int32)
        nb = C.c_int()
        _check(lib.mi355lz4_index_host(src.ctypes.data_as(_u8p), src.size, header_kind, fixed_uncomp,
                                       boff.ctypes.data_as(_u64p), ulen.ctypes.data_as(_i32p), max_blocks, C.byref(nb)),
               "index_host")
        cap = int(ulen[: nb.value].astype(np.int64).clip(min=0).sum()) + 16
        out = np.empty(cap, dtype=np.uint8)
        out_len = C.c_size_t()
        blen = np.zeros(max(nb.value, 1), dtype=np.int32)
        got = C.c_int()
        d = np.frombuffer(bytes(dict_bytes), dtype=np.uint8) if dict_bytes else None
        rc = lib.mi355lz4_decompress_batch(self.ctx, src.ctypes.data_as(_u8p), src.size, header_kind, fixed_uncomp,
                                           int(bool(linked)), d.ctypes.data_as(_u8p) if d is not None else None,
                                           d.size if d is not None else 0, out.ctypes.data_as(_u8p), cap,
                                           C.byref(out_len), blen.ctypes.data_as(_i32p), max(nb.value, 1), C.byref(got))
        if rc != 0 and (raise_on_block_error or rc != -5):
            _check(rc, "decompress_batch")
        return out[: out_len.value].tobytes(), blen[: got.value].tolist()

    def decompress_streams(self, framed, stream_first, header_kind=8, fixed_uncomp=0, raise_on_block_error=True):
        """Many linked streams, host buffers: stream s = blocks [stream_first[s], stream_first[s+1]).
        Returns (decoded bytes, [decoded length or negative code per block])."""
        src = np.frombuffer(bytes(framed), dtype=np.uint8)
        max_blocks = src.size // (header_kind + 1) + 1
        boff = np.zeros(max_blocks + 1, dtype=np.uint64)
        ulen = np.zeros(max_blocks + 1, dtype=np.int32)
        nb = C.c_int()
        _check(lib.mi355lz4_index_host(src.ctypes.data_as(_u8p), src.size, header_kind, fixed_uncomp,
                                       boff.ctypes.data_as(_u64p), ulen.ctypes.data_as(_i32p), max_blocks, C.byref(nb)),
               "index_host")
        cap = int(ulen[: nb.value].astype(np.int64).clip(min=0).sum()) + 16
        out = np.empty(cap, dtype=np.uint8)
        out_len = C.c_size_t()
        blen = np.zeros(max(nb.value, 1), dtype=np.int32)
        got = C.c_int()
        sf = np.asarray(stream_first, dtype=np.int32)
        rc = lib.mi355lz4_decompress_streams(self.ctx, src.ctypes.data_as(_u8p), src.size, header_kind, fixed_uncomp,
                                             sf.ctypes.data_as(_i32p), int(sf.size - 1), out.ctypes.data_as(_u8p), cap,
                                             C.byref(out_len), blen.ctypes.data_as(_i32p), max(nb.value, 1), C.byref(got))
        if rc != 0 and (raise_on_block_error or rc != -5):
            _check(rc, "decompress_streams")
        return out[: out_len.value].tobytes(), blen[: got.value].tolist()


# ---------------------------------------------------------------------------
# Stream-combinator mirror (lists / iterables of bytes in, list of bytes out)
# ---------------------------------------------------------------------------
def _pack(arrays):
    arrays = [bytes(a) for a in arrays]
    data = np.frombuffer(b"".join(arrays), dtype=np.uint8) if arrays else np.zeros(0, dtype=np.uint8)
    if data.size == 0:
        data = np.zeros(1, dtype=np.uint8)
    lens = np.array([len(a) for a in arrays] + [0], dtype=np.uint64)
    return data, lens, len(arrays)


class _ArraysOwner:
    """Keeps a C-side result alive while Python slices of it are in use."""
    def __init__(self, h):
        self.h = h

    def __del__(self):
        if self.h and lib is not None:                 # (module globals are gone at interpreter shutdown)
            lib.slz4_arrays_free(self.h)
            self.h = None


def _unpack_views(h):
    """The result arrays as memoryviews INTO the C-side buffer (no copy): what a Haskell `Array` is -- a slice of a
    shared buffer.  The buffer lives as long as any of the views."""
    owner = _ArraysOwner(h)
    n = lib.slz4_arrays_count(h)
    base, offs = _u8p(), C.POINTER(C.c_size_t)()
    if n and lib.slz4_arrays_flat(h, C.byref(base), C.byref(offs)):
        # one buffer: one ctypes object, the arrays are slices of its memoryview
        total = offs[n]
        if total == 0:
            return [memoryview(b"")] * n
        arr = (C.c_uint8 * total).from_address(C.cast(base, C.c_void_p).value)
        arr._owner = owner
        mv = memoryview(arr).cast("B")
        o = offs[0:n + 1]
        return [mv[o[i]:o[i + 1]] for i in range(n)]
    out = []
    for i in range(n):
        ln = lib.slz4_arrays_len(h, i)
        if ln:
            arr = (C.c_uint8 * ln).from_address(C.cast(lib.slz4_arrays_data(h, i), C.c_void_p).value)
            arr._owner = owner
            out.append(memoryview(arr).cast("B"))
        else:
            out.append(memoryview(b""))
    return out


def _unpack(h):
    try:
        n = lib.slz4_arrays_count(h)
        out = []
        for i in range(n):
            ln = lib.slz4_arrays_len(h, i)
            out.append(C.string_at(lib.slz4_arrays_data(h, i), ln) if ln else b"")
        return out
    finally:
        lib.slz4_arrays_free(h)


def _run(fn, *args, views=False):
    h = C.c_void_p()
    rc = fn(*args, C.byref(h))
    if rc != 0:
        raise LZ4Error((lib.slz4_last_error() or b"").decode("utf-8", "replace"))
    return _unpack_views(h) if views else _unpack(h)


def compressChunks(cfg, speed, arrays, engine):
    """Streamly.LZ4.compressChunks (reference src/Streamly/LZ4.hs:94-100)."""
    data, lens, n = _pack(arrays)
    return _run(lib.slz4_compress_chunks, engine._h, cfg.blockSize, int(speed), data.ctypes.data_as(_u8p),
                lens.ctypes.data_as(_u64p), n)


def resizeChunks(cfg, conf, arrays):
    """Streamly.Internal.LZ4.resizeChunksD (reference src/Streamly/Internal/LZ4.hs:432-523).  Host only."""
    data, lens, n = _pack(arrays)
    return _run(lib.slz4_resize_chunks, cfg.blockSize, int(conf.hasEndMark), data.ctypes.data_as(_u8p),
                lens.ctypes.data_as(_u64p), n)


def decompressChunksRaw(cfg, arrays, engine):
    """Streamly.Internal.LZ4.decompressChunksRawD (reference src/Streamly/Internal/LZ4.hs:539-567)."""
    data, lens, n = _pack(arrays)
    return _run(lib.slz4_decompress_chunks_raw, engine._h, cfg.blockSize, data.ctypes.data_as(_u8p),
                lens.ctypes.data_as(_u64p), n)


def decompressChunks(cfg, arrays, engine, conf=defaultFrameConfig, views=False):
    """Streamly.LZ4.decompressChunks (reference src/Streamly/LZ4.hs:114-122); conf selects the end-mark variant.
    views=True returns the arrays as memoryviews into one result buffer (the reference's arrays are such slices) instead
    of one bytes object -- one copy -- per array."""
    data, lens, n = _pack(arrays)
    return _run(lib.slz4_decompress_chunks, engine._h, cfg.blockSize, int(conf.hasEndMark), data.ctypes.data_as(_u8p),
                lens.ctypes.data_as(_u64p), n, views=views)


def decompressChunksWith(arrays, engine):
    """decompressChunksWithD simpleFrameParserD (reference src/Streamly/Internal/LZ4.hs:569-577)."""
    data, lens, n = _pack(arrays)
    return _run(lib.slz4_decompress_chunks_with, engine._h, data.ctypes.data_as(_u8p), lens.ctypes.data_as(_u64p), n)


def simpleFrameParser(arrays):
    """simpleFrameParserD (reference src/Streamly/Internal/LZ4.hs:590-651).  Returns ((BlockConfig, FrameConfig), rest)."""
    data, lens, n = _pack(arrays)
    h = C.c_void_p()
    em = C.c_int()
    kind = lib.slz4_simple_frame_parser(data.ctypes.data_as(_u8p), lens.ctypes.data_as(_u64p), n, C.byref(em), C.byref(h))
    if kind < 0:
        raise LZ4Error((lib.slz4_last_error() or b"").decode("utf-8", "replace"))
    return (BlockConfig(kind), FrameConfig(bool(em.value))), _unpack(h)


# ---- the standard LZ4 frame format (csrc/lz4_frame.cpp): interop with liblz4's LZ4F_* and the lz4 CLI ----
def _bytes_arg(data):
    a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
    if a.size == 0:
        return np.zeros(1, dtype=np.uint8), 0
    return a, int(a.size)


def xxh32(data, seed=0):
    """xxHash32, the checksum of the LZ4 frame format (host code, no GPU)."""
    a, n = _bytes_arg(data)
    return int(lib.slz4_xxh32(a.ctypes.data_as(_u8p), n, int(seed)))


def lz4FrameCompress(data, engine, speed=1, blockMax=BlockSize.BlockMax64KB, blockChecksum=False, contentChecksum=True,
                     contentSize=False, linkedBlocks=False):
    """One standard LZ4 frame (independent blocks, or linked ones: smaller on text) -- readable by LZ4F_decompress / `lz4 -d`.  The reference's
    own frame support stops at parsing a header without these options (src/Streamly/Internal/LZ4.hs:631-638)."""
    a, n = _bytes_arg(data)
    flags = (1 if blockChecksum else 0) | (2 if contentChecksum else 0) | (4 if contentSize else 0) | (8 if linkedBlocks else 0)
    return _run(lib.slz4_lz4frame_compress, engine._h, int(blockMax), flags, int(speed), a.ctypes.data_as(_u8p), n)[0]


def lz4FrameDecompress(frame, engine):
    """Decode any sequence of standard LZ4 frames (linked or independent blocks, stored blocks, checksums verified)."""
    a, n = _bytes_arg(frame)
    return _run(lib.slz4_lz4frame_decompress, engine._h, a.ctypes.data_as(_u8p), n)[0]
