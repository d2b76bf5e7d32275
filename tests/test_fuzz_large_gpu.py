"""Extended decoder fuzz on full-size blocks (the lane-parallel path deep inside a block): thousands of
blocks, each with its own corruption (flipped bits, overwritten runs, truncation, wrong capacity), decoded
in ONE batch; every result code and every decoded byte must equal the oracle's (reference semantics,
cbits/lz4.c:1737-2165).  FUZZ_CASES raises the count for soak runs."""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("decoder", [2, 1, 4])
def test_fuzz_large_blocks(engine, oracle, decoder):
    import torch
    dev = torch.device("cuda:0")
    n_cases = int(os.environ.get("FUZZ_CASES", "1500"))
    rng = random.Random(99 + decoder)
    base = {}
    for kind in ("lzsynth", "text"):
        for bl in (4096, 20000, 65536):
            d = oracle.gen(kind, 1, bl, first_block=7).tobytes()
            base[(kind, bl)] = (d, oracle.compress_block(d, 1))
    keys = sorted(base)
    payloads, caps = [], []
    for t in range(n_cases):
        d, comp = base[keys[t % len(keys)]]
        comp = bytearray(comp)
        cap = len(d)
        mode = rng.randrange(6)
        if mode == 0:                                   # single bit
            comp[rng.randrange(len(comp))] ^= 1 << rng.randrange(8)
        elif mode == 1:                                 # a few random bytes
            for _ in range(rng.randrange(1, 5)):
                comp[rng.randrange(len(comp))] = rng.randrange(256)
        elif mode == 2:                                 # truncation
            comp = comp[: rng.randrange(1, len(comp))]
        elif mode == 3:                                 # a run of 0xFF (length-extension storms) or zeros (offset 0)
            p = rng.randrange(len(comp))
            comp[p:p + rng.randrange(1, 40)] = bytes([rng.choice([0xFF, 0x00])]) * min(40, len(comp) - p)
            comp = comp[: max(1, len(comp))]
        elif mode == 4:                                 # capacity too small / too large
            cap = max(0, cap + rng.choice([-1, -7, -64, -1000, 5, 300]))
        # mode 5: untouched
        payloads.append(bytes(comp))
        caps.append(cap)
    # one framed batch, headerKind 8 (every block carries its own capacity)
    blob = bytearray()
    boff = []
    for p, cap in zip(payloads, caps):
        boff.append(len(blob))
        blob += len(p).to_bytes(4, "little") + cap.to_bytes(4, "little") + p
    blob += bytes(64)                                   # out-of-block reads of malformed input see zeros, as in the oracle's padded buffers
    ooff = np.concatenate([[0], np.cumsum([c + 64 for c in caps])]).astype(np.int64)
    buf = torch.from_numpy(np.frombuffer(bytes(blob), dtype=np.uint8).copy()).to(dev)
    off = torch.tensor(boff, dtype=torch.int64, device=dev)
    oo = torch.from_numpy(ooff).to(dev)
    out = torch.zeros(int(ooff[-1]) + 64, dtype=torch.uint8, device=dev)
    res = torch.zeros(n_cases, dtype=torch.int32, device=dev)
    engine.set_decoder(decoder)
    try:
        engine.decompress_batch_device(buf, len(blob) - 64, off, n_cases, out, oo, res)
        engine.synchronize()
    finally:
        engine.set_decoder(0)
    got = res.cpu().tolist()
    outh = out.cpu().numpy()
    bad = 0
    for i, (p, cap) in enumerate(zip(payloads, caps)):
        code, dec = oracle.decompress_block(p, cap)
        assert got[i] == code, (i, len(p), cap, got[i], code)
        if code >= 0:
            assert outh[ooff[i]:ooff[i] + code].tobytes() == dec, i
        else:
            bad += 1
    assert bad > n_cases // 4                           # the corruptions really bite
