"""Development aid: what the asynchronous linked decode (mi355lz4_set_linked_async) costs / saves against the default
(one host wait): independent blocks with linked = 1, and a reference-written linked text stream."""
import os, struct, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import numpy as np, torch, streamly_lz4_amd as S
from oracle.oracle import Oracle, Reference, have_reference
O = Reference() if have_reference() else Oracle()
eng = S.Engine(0); BL = 65536; dev = "cuda"
def timed(fn, n=5):
    e0, e1 = S.Event(), S.Event(); best = 1e9
    for _ in range(n):
        eng.record(e0); fn(); eng.record(e1); eng.synchronize(); best = min(best, eng.elapsed_ms(e0, e1))
    return best
for NB in (160, 16384):
    src = torch.empty(NB * BL, dtype=torch.uint8, device=dev); eng.generate("text", src, BL, NB)
    stride = S.slot_stride(BL, 8)
    slots = torch.empty(NB * stride, dtype=torch.uint8, device=dev); flen = torch.empty(NB, dtype=torch.int32, device=dev)
    dense = torch.empty(NB * stride, dtype=torch.uint8, device=dev); doff = torch.empty(NB + 1, dtype=torch.int64, device=dev)
    eng.compress_batch_device(src, NB, BL, slots, stride, flen); eng.compact_device(slots, stride, flen, NB, dense, NB * stride, doff); eng.synchronize()
    cb = int(doff[-1].item())
    out = torch.zeros(NB * BL, dtype=torch.uint8, device=dev); res = torch.zeros(NB, dtype=torch.int32, device=dev)
    ooff = torch.arange(NB + 1, dtype=torch.int64, device=dev) * BL
    row = {"blocks": NB, "independent_blocks": {}}
    for name, lk, cap in (("linked0", False, 0), ("linked1_wait", True, 0), ("linked1_async", True, BL)):
        eng.set_linked_async(cap)
        row["independent_blocks"][name + "_ms"] = round(timed(lambda: eng.decompress_batch_device(dense, cb, doff, NB, out, ooff, res, linked=lk)), 4)
    eng.set_linked_async(0)
    assert torch.equal(out, src)
    if NB <= 4096 * 4:
        nl = min(NB, 4096)
        raw = src[: nl * BL].cpu().numpy().tobytes()
        framed = O.frame_compress(raw, BL, 1, 8, True)
        offs, pos = [], 0
        while pos < len(framed):
            offs.append(pos); pos += 8 + struct.unpack_from("<i", framed, pos)[0]
        offs.append(pos)
        fr = torch.from_numpy(np.frombuffer(framed, dtype=np.uint8).copy()).cuda(); boff = torch.tensor(offs, dtype=torch.int64).cuda()
        row["reference_linked_stream"] = {"blocks": nl}
        for name, cap in (("wait", 0), ("async", BL)):
            eng.set_linked_async(cap)
            row["reference_linked_stream"][name + "_ms"] = round(timed(lambda: eng.decompress_batch_device(fr, len(framed), boff, nl, out, ooff, res, linked=True)), 4)
        eng.set_linked_async(0)
        assert torch.equal(out[: nl * BL], src[: nl * BL])
    print(row, flush=True)
