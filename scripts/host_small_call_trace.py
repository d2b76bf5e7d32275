"""Development aid: a few 10 MiB host-buffer decompress calls through the C ABI (for a rocprofv3 --hip-trace --kernel-trace
--memory-copy-trace timeline: where the call's 0.6 ms go).   python scripts/host_small_call_trace.py [calls]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import numpy as np, torch, ctypes as C, streamly_lz4_amd as S
eng = S.Engine(0)
N = 10 << 20; BL = 65536; calls = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 12
t = torch.empty(N, dtype=torch.uint8, device="cuda:0"); eng.generate("lzsynth", t, BL, N // BL); eng.synchronize()
raw = t.cpu().numpy().tobytes()
framed, _ = eng.compress_batch([raw[i:i + BL] for i in range(0, N, BL)])
src = np.frombuffer(framed, dtype=np.uint8); outb = np.empty(N + 16, dtype=np.uint8)
if "pinned" in sys.argv:            # page-locked caller memory: no staging copies either way
    ps = torch.empty(len(framed), dtype=torch.uint8).pin_memory(); ps.numpy()[:] = src; src = ps.numpy()
    po = torch.empty(N + 16, dtype=torch.uint8).pin_memory(); outb = po.numpy()
blen = np.zeros(N // BL + 1, dtype=np.int32); ol = C.c_size_t(); got = C.c_int()
best = 1e9
for _ in range(calls):
    t0 = time.perf_counter()
    rc = S.lib.mi355lz4_decompress_batch(eng.ctx, src.ctypes.data_as(S._u8p), src.size, 8, 0, 1, None, 0, outb.ctypes.data_as(S._u8p), N + 16,
                                         C.byref(ol), blen.ctypes.data_as(S._i32p), N // BL, C.byref(got))
    best = min(best, time.perf_counter() - t0)
    assert rc == 0
assert outb[:N].tobytes() == raw
print("c_abi_host_ms", "pinned" if "pinned" in sys.argv else "pageable", round(best * 1e3, 4), "framed_MB", round(len(framed) / 1e6, 2))
