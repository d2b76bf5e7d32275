"""ctypes bindings for the CPU oracle (liboracle.so) and, when built, the real
reference codec (oracle/_ref/liblz4ref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PAD = 64  # readable slack after every buffer handed to a decoder (malformed-input probes)

_u8p = C.POINTER(C.c_uint8)


def build(force=False):
    """Compile liboracle.so (and _ref when /root/reference is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    if force or not os.path.exists(so) or os.path.exists("/root/reference/cbits/lz4.c") and not os.path.exists(
        os.path.join(_HERE, "_ref", "liblz4ref.so")
    ):
        subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
    return so


def _ptr(a):
    return a.ctypes.data_as(_u8p)


def _padded(buf):
    """uint8 copy of buf with _PAD zero bytes of readable slack behind it."""
    b = np.frombuffer(bytes(buf), dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf.astype(np.uint8, copy=False).ravel()
    out = np.zeros(b.size + _PAD, dtype=np.uint8)
    out[: b.size] = b
    return out, b.size


class _Lib:
    def __init__(self, path, prefix):
        self.lib = C.CDLL(path, mode=os.RTLD_LOCAL)
        self.prefix = prefix
        L = self.lib
        f = getattr(L, prefix + "_frame_stream_compress")
        f.restype = C.c_size_t
        f.argtypes = [_u8p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, C.c_size_t]
        f = getattr(L, prefix + "_frame_stream_decompress")
        f.restype = C.c_size_t
        f.argtypes = [_u8p, C.c_size_t, C.c_int, C.c_int, C.c_int, _u8p, C.c_size_t]
        f = getattr(L, prefix + "_compress_block")
        f.restype = C.c_int
        f.argtypes = [_u8p, _u8p, C.c_int, C.c_int, C.c_int]
        for name in ("_time_compress", "_time_decompress"):
            getattr(L, prefix + name).restype = C.c_double

    # -- one independent block through a fresh context -------------------
    def compress_block(self, data, accel=1):
        src, n = _padded(data)
        bound = n + n // 255 + 16
        dst = np.zeros(bound + _PAD, dtype=np.uint8)
        r = getattr(self.lib, self.prefix + "_compress_block")(_ptr(src), _ptr(dst), n, bound, accel)
        if r <= 0:
            raise RuntimeError("compress_block failed: %d" % r)
        return dst[:r].tobytes()

    # -- framed streams ---------------------------------------------------
    def frame_compress(self, data, block_len=65536, accel=1, header=8, linked=True):
        src, n = _padded(data)
        nblk = max(1, (n + block_len - 1) // block_len)
        cap = n + n // 255 + (16 + header) * nblk + 64
        dst = np.zeros(cap, dtype=np.uint8)
        r = getattr(self.lib, self.prefix + "_frame_stream_compress")(
            _ptr(src), n, block_len, accel, header, int(linked), _ptr(dst), cap
        )
        if r == C.c_size_t(-1).value:
            raise RuntimeError("frame_compress failed")
        return dst[:r].tobytes()

    def frame_decompress(self, framed, out_cap, header=8, fixed_uncomp=65536, linked=True):
        src, n = _padded(framed)
        dst = np.zeros(out_cap + _PAD, dtype=np.uint8)
        r = getattr(self.lib, self.prefix + "_frame_stream_decompress")(
            _ptr(src), n, header, fixed_uncomp, int(linked), _ptr(dst), out_cap
        )
        if r > out_cap:
            k = C.c_size_t(-1).value - r
            raise RuntimeError("frame_decompress failed at block %d" % k)
        return dst[:r].tobytes()


class Oracle(_Lib):
    def __init__(self):
        super().__init__(build(), "orc")
        L = self.lib
        L.orc_decompress_safe_dict.restype = C.c_int
        L.orc_decompress_safe_dict.argtypes = [_u8p, C.c_int, _u8p, C.c_int, _u8p, C.c_size_t]
        L.orc_compress_bound.restype = C.c_int
        L.orc_compress_bound.argtypes = [C.c_int]
        L.orc_gen_random.argtypes = [_u8p, C.c_size_t, C.c_uint64]
        L.orc_gen_lzsynth.argtypes = [_u8p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_uint32]
        L.orc_gen_text.argtypes = [_u8p, C.c_size_t, C.c_uint64]

    def compress_bound(self, n):
        return self.lib.orc_compress_bound(n)

    def decompress_block(self, comp, cap, dict_bytes=None):
        """Returns (code, bytes).  code >= 0 is the decoded size, < 0 the reference error code."""
        src, n = _padded(comp)
        dst = np.zeros(cap + _PAD, dtype=np.uint8)
        if dict_bytes:
            d, dn = _padded(dict_bytes)
            r = self.lib.orc_decompress_safe_dict(_ptr(src), n, _ptr(dst), cap, _ptr(d), dn)
        else:
            r = self.lib.orc_decompress_safe_dict(_ptr(src), n, _ptr(dst), cap, None, 0)
        return r, (dst[:r].tobytes() if r >= 0 else b"")

    # -- generators -------------------------------------------------------
    def gen(self, kind, n_blocks, block_len, first_block=0, lit_max=16, off_max=2048):
        out = np.empty(n_blocks * block_len, dtype=np.uint8)
        for i in range(n_blocks):
            p = out[i * block_len:].ctypes.data_as(_u8p)
            if kind == "random":
                self.lib.orc_gen_random(p, block_len, first_block + i)
            elif kind == "lzsynth":
                self.lib.orc_gen_lzsynth(p, block_len, first_block + i, lit_max, off_max)
            elif kind == "text":
                self.lib.orc_gen_text(p, block_len, first_block + i)
            else:
                raise ValueError(kind)
        return out


class Reference(_Lib):
    """The real reference codec (cbits/lz4.c) -- only where oracle/_ref was built."""

    def __init__(self):
        build()
        path = os.path.join(_HERE, "_ref", "liblz4ref.so")
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        super().__init__(path, "ref")
        L = self.lib
        L.ref_decompress_block_dict.restype = C.c_int
        L.ref_decompress_block_dict.argtypes = [_u8p, C.c_int, _u8p, C.c_int, _u8p, C.c_int]
        L.ref_version.restype = C.c_int

    def version(self):
        return self.lib.ref_version()

    def decompress_block(self, comp, cap, dict_bytes=None):
        src, n = _padded(comp)
        dst = np.zeros(cap + _PAD, dtype=np.uint8)
        if dict_bytes:
            d, dn = _padded(dict_bytes)
            r = self.lib.ref_decompress_block_dict(_ptr(src), n, _ptr(dst), cap, _ptr(d), dn)
        else:
            r = self.lib.ref_decompress_block_dict(_ptr(src), n, _ptr(dst), cap, None, 0)
        return r, (dst[:r].tobytes() if r >= 0 else b"")


def have_reference():
    return os.path.exists(os.path.join(_HERE, "_ref", "liblz4ref.so")) or os.path.exists("/root/reference/cbits/lz4.c")


def cpu_share():
    """Host threads this process may really use: the cgroup CPU quota when one is set, else the CPU
    count capped at 16 (a 1-GPU box's share of its host)."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return max(1, int(int(q) / int(per)))
    except Exception:
        pass
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    return max(1, min(n, 16))


class _CpuJob:
    """Buffers + timed calls for one host thread's share of the CPU baseline."""

    def __init__(self, blocks, accel):
        if have_reference():
            self.lib, self.prefix, self.kind = Reference().lib, "ref", "reference"
        else:
            self.lib, self.prefix, self.kind = Oracle().lib, "orc", "port"
        self.accel = accel
        n = self.n = len(blocks)
        self.ins = [np.frombuffer(b, dtype=np.uint8).copy() for b in blocks]
        self.lens = np.array([a.size for a in self.ins], dtype=np.int32)
        bounds = [int(a.size + a.size // 255 + 16) for a in self.ins]
        self.comps = [np.zeros(b + _PAD, dtype=np.uint8) for b in bounds]
        self.outs = [np.zeros(a.size + _PAD, dtype=np.uint8) for a in self.ins]
        self.clens = np.zeros(n, dtype=np.int32)
        self.res = np.zeros(n, dtype=np.int32)
        PP = _u8p * n
        self.in_pp = PP(*[_ptr(a) for a in self.ins])
        self.comp_pp = PP(*[_ptr(a) for a in self.comps])
        self.out_pp = PP(*[_ptr(a) for a in self.outs])

    def compress(self):
        ip = C.POINTER(C.c_int)
        return getattr(self.lib, self.prefix + "_time_compress")(
            self.in_pp, self.lens.ctypes.data_as(ip), self.n, self.accel, self.comp_pp, self.clens.ctypes.data_as(ip))

    def decompress(self):
        ip = C.POINTER(C.c_int)
        return getattr(self.lib, self.prefix + "_time_decompress")(
            self.comp_pp, self.clens.ctypes.data_as(ip), self.n, self.out_pp, self.lens.ctypes.data_as(ip),
            self.res.ctypes.data_as(ip))

    def verify(self):
        for a, o, r in zip(self.ins, self.outs, self.res):
            if r != a.size or not np.array_equal(o[: a.size], a):
                raise RuntimeError("cpu baseline round trip mismatch")


def cpu_baseline_all_cores(blocks, accel=1, threads=None, reps=2):
    """Best-case CPU (NOT the reference's behaviour, whose API is serial): `threads` host threads, one
    independent linked context per thread over a contiguous range of `blocks`.  All threads start each
    timed phase together (barrier); the phase time is the wall time from the barrier to the last
    thread's return.  Used only by bench.py's cpu_baseline leg."""
    import threading
    import time
    threads = threads or cpu_share()
    threads = max(1, min(threads, len(blocks)))
    per = (len(blocks) + threads - 1) // threads
    parts = [blocks[i:i + per] for i in range(0, len(blocks), per)]
    jobs = [_CpuJob(p, accel) for p in parts]
    nt = len(jobs)

    def phase(fn_name):
        bar = threading.Barrier(nt + 1)
        ends = [0.0] * nt

        def run(i):
            fn = getattr(jobs[i], fn_name)
            bar.wait()
            fn()
            ends[i] = time.perf_counter()

        ts = [threading.Thread(target=run, args=(i,)) for i in range(nt)]
        for t in ts:
            t.start()
        bar.wait()
        t0 = time.perf_counter()
        for t in ts:
            t.join()
        return max(ends) - t0

    comp_s = min(phase("compress") for _ in range(reps))
    decomp_s = min(phase("decompress") for _ in range(reps))
    for j in jobs:
        j.verify()
    return {"kind": jobs[0].kind, "threads": nt, "comp_s": comp_s, "decomp_s": decomp_s,
            "raw_bytes": int(sum(int(j.lens.sum()) for j in jobs)), "comp_bytes": int(sum(int(j.clens.sum()) for j in jobs))}


def cpu_baseline(blocks, accel=1, reps=3, keep_stream=False):
    """Time compress+decompress of `blocks` (list of bytes) with the reference call
    sequence on ONE host thread.  Returns dict(kind, comp_s, decomp_s, comp_bytes).
    keep_stream: also return the compressed blocks ("stream": list of bytes) -- the linked stream the
    reference writes for this input, which bench.py hands to the GPU decoder.
    Used only by bench.py's cpu_baseline leg."""
    job = _CpuJob(blocks, accel)
    best_c = min(job.compress() for _ in range(reps))
    best_d = min(job.decompress() for _ in range(reps))
    job.verify()
    r = {"kind": job.kind, "comp_s": best_c, "decomp_s": best_d, "comp_bytes": int(job.clens.sum()),
         "raw_bytes": int(job.lens.sum())}
    if keep_stream:
        r["stream"] = [c[:int(n)].tobytes() for c, n in zip(job.comps, job.clens)]
    return r
