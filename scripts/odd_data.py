"""Development aid: throughput on awkward inputs (long runs, tiny alphabets) to catch pathological paths."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import numpy as np, torch
import streamly_lz4_amd as S
dev = torch.device("cuda:0"); eng = S.Engine(0); BL = 65536; NB = 16384
rng = np.random.default_rng(1)
cases = {
    "zeros": np.zeros(NB * BL, np.uint8),
    "bits01": rng.integers(0, 2, NB * BL, dtype=np.uint8),                         # the reference tests' generator
    "hc_90_10": (rng.random(NB * BL) < 0.1).astype(np.uint8),                       # genArrayW8LargeHC
    "period7": np.tile(np.arange(7, dtype=np.uint8), NB * BL // 7 + 1)[: NB * BL],
    "period300": np.tile(rng.integers(0, 256, 300, dtype=np.uint8), NB * BL // 300 + 1)[: NB * BL],
    "runs": np.repeat(rng.integers(0, 256, NB * BL // 500 + 1, dtype=np.uint8), 500)[: NB * BL],
}
for name, host in cases.items():
    src = torch.from_numpy(host).to(dev)
    stride = S.slot_stride(BL, 8)
    slots = torch.empty(NB * stride, dtype=torch.uint8, device=dev); flen = torch.empty(NB, dtype=torch.int32, device=dev)
    dense = torch.empty(NB * stride, dtype=torch.uint8, device=dev); doff = torch.empty(NB + 1, dtype=torch.int64, device=dev)
    ooff = torch.arange(NB + 1, dtype=torch.int64, device=dev) * BL
    out = torch.empty(NB * BL, dtype=torch.uint8, device=dev); res = torch.empty(NB, dtype=torch.int32, device=dev)
    e = [S.Event() for _ in range(3)]
    tc = td = 1e9
    for it in range(3):
        eng.record(e[0]); eng.compress_batch_device(src, NB, BL, slots, stride, flen); eng.record(e[1])
        eng.compact_device(slots, stride, flen, NB, dense, NB * stride, doff); eng.synchronize()
        tc = min(tc, eng.elapsed_ms(e[0], e[1]))
        eng.record(e[1]); eng.decompress_batch_device(dense, NB * stride, doff, NB, out, ooff, res); eng.record(e[2]); eng.synchronize()
        td = min(td, eng.elapsed_ms(e[1], e[2]))
    ok = bool((res == BL).all().item()) and torch.equal(out, src)
    C = int(doff[-1].item()); U = NB * BL
    print("%-10s ratio %8.2f  enc %7.1f GB/s  dec %7.1f GB/s  ok=%s" % (name, U / C, U / tc / 1e6, U / td / 1e6, ok))
