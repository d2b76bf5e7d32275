"""Development aid: do encoder launches on different streams run side by side?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import torch, streamly_lz4_amd as S
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
dev = torch.device("cuda:0"); eng = S.Engine(0)
BL, NB = 65536, 8192
src = torch.empty(NB * BL, dtype=torch.uint8, device=dev); eng.generate("lzsynth", src, BL, NB)
stride = S.slot_stride(BL, 8)
slots = torch.empty(NB * stride, dtype=torch.uint8, device=dev); flen = torch.empty(NB, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
for nstreams in (1, 2, 4, 8):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    for per in (512, 1024):
        ng = NB // per
        def run():
            for g in range(ng):
                with torch.cuda.stream(streams[g % nstreams]):
                    eng.compress_batch_device(src[g * per * BL:], per, BL, slots[g * per * stride:], stride, flen[g * per:])
        run(); torch.cuda.synchronize()
        t0 = time.perf_counter(); run(); torch.cuda.synchronize(); t1 = time.perf_counter()
        print("%d stream(s), %d launches of %d blocks: %.2f ms" % (nstreams, ng, per, (t1 - t0) * 1e3))
t0 = time.perf_counter(); eng.compress_batch_device(src, NB, BL, slots, stride, flen); torch.cuda.synchronize()
print("one launch of %d blocks: %.2f ms" % (NB, (time.perf_counter() - t0) * 1e3))
