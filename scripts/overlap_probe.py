"""Development aid: does a pinned H2D / D2H copy on a side stream overlap with the codec kernels, and at what rate?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import torch, streamly_lz4_amd as S
print("HSA_ENABLE_SDMA =", os.environ.get("HSA_ENABLE_SDMA"))
dev = torch.device("cuda:0"); eng = S.Engine(0)
BL, NB = 65536, 8192
src = torch.empty(NB * BL, dtype=torch.uint8, device=dev); eng.generate("lzsynth", src, BL, NB)
stride = S.slot_stride(BL, 8)
slots = torch.empty(NB * stride, dtype=torch.uint8, device=dev); flen = torch.empty(NB, dtype=torch.int32, device=dev)
n = 512 << 20
h = torch.empty(n, dtype=torch.uint8).pin_memory(); d = torch.empty(n, dtype=torch.uint8, device=dev)
side = torch.cuda.Stream()
def copy_chunks(direction, chunk):
    with torch.cuda.stream(side):
        for o in range(0, n, chunk):
            if direction == "H2D": d[o:o + chunk].copy_(h[o:o + chunk], non_blocking=True)
            else: h[o:o + chunk].copy_(d[o:o + chunk], non_blocking=True)
for direction in ("H2D", "D2H"):
    for chunk in (64 << 20, 512 << 20):
        copy_chunks(direction, chunk); torch.cuda.synchronize()
        t0 = time.perf_counter(); copy_chunks(direction, chunk); side.synchronize(); t1 = time.perf_counter()
        print("%s alone, %d MiB chunks: %.1f GB/s" % (direction, chunk >> 20, n / (t1 - t0) / 1e9))
    # with the encoder busy on the main stream
    for _ in range(2): eng.compress_batch_device(src, NB, BL, slots, stride, flen)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(4): eng.compress_batch_device(src, NB, BL, slots, stride, flen)     # ~20 ms of kernel
    copy_chunks(direction, 64 << 20); side.synchronize(); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    print("%s beside 4 encode launches: copy done after %.2f ms (%.1f GB/s), kernels done after %.2f ms" % (direction, (t1 - t0) * 1e3, n / (t1 - t0) / 1e9, (t2 - t0) * 1e3))
t0 = time.perf_counter()
for _ in range(4): eng.compress_batch_device(src, NB, BL, slots, stride, flen)
torch.cuda.synchronize(); print("4 encode launches alone: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
