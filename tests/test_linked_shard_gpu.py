"""ONE reference-written (linked) stream decoded by several ranks, contiguous block ranges, the seam block passed from
rank to rank (streamly_lz4_amd/linked_shard.py; reference cbits/lz4.c:2347-2355).

Rehearsed the way bench.py rehearses its N > 1 path on a one-GPU box: every rank on device 0, gloo rendezvous.  The
result of every rank must equal the oracle's linked decode (= the input) for its range; the C ABI's two-call form
must give what the one-call linked decode gives."""
import os
import struct
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BL = 65536


def _stream(oracle, kind, n_blocks, seed):
    """(raw bytes, framed linked stream, block header offsets)"""
    if kind == "shared":
        rng = np.random.default_rng(seed)
        vocab = [bytes(rng.integers(97, 123, size=int(k), dtype=np.uint8)) for k in rng.integers(3, 12, size=300)]
        words = rng.integers(0, len(vocab), size=n_blocks * BL // 4)
        raw = b" ".join(vocab[i] for i in words)[: n_blocks * BL]
    else:
        raw = oracle.gen(kind, n_blocks, BL, first_block=seed).tobytes()
    framed = oracle.frame_compress(raw, BL, 1, 8, True)
    offs, pos = [], 0
    while pos < len(framed):
        offs.append(pos)
        pos += 8 + struct.unpack_from("<i", framed, pos)[0]
    offs.append(pos)
    assert len(offs) == n_blocks + 1
    return raw, framed, offs


def _worker(rank, world, port, kind, n_blocks, cuts, q, early=True, env=None):
    for p in (ROOT, os.path.join(ROOT, "streamly-lz4_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.update(env or {})
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import streamly_lz4_amd as S
        from streamly_lz4_amd.linked_shard import decode_linked_sharded
        from oracle.oracle import Oracle
        raw, framed, offs = _stream(Oracle(), kind, n_blocks, 5)
        b0, b1 = cuts[rank], cuts[rank + 1]
        torch.cuda.set_device(0)
        eng = S.Engine(0)
        mine = framed[offs[b0]:offs[b1]]
        fr = torch.from_numpy(np.frombuffer(mine, dtype=np.uint8).copy()).cuda()
        boff = torch.tensor([o - offs[b0] for o in offs[b0:b1 + 1]], dtype=torch.int64).cuda()
        out, res = decode_linked_sharded(eng, fr, len(mine), boff, [BL] * (b1 - b0), use_end_last=early)
        ok = res.cpu().tolist() == [BL] * (b1 - b0) and out.cpu().numpy().tobytes() == raw[b0 * BL:b1 * BL]
        q.put((rank, bool(ok)))
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind,n_blocks,cuts,early", [("text", 48, [0, 20, 48], True), ("shared", 40, [0, 1, 40], True),
                                                      ("text", 96, [0, 30, 61, 96], True), ("lzsynth", 16, [0, 8, 16], True),
                                                      ("text", 56, [0, 30, 56], False),
                                                      ("text", 44, [0, 21, 44], "segments"),
                                                      ("text", 52, [0, 27, 52], "runin")])
def test_one_linked_stream_over_ranks(kind, n_blocks, cuts, early):
    """early: a rank hands its last block on before the rest of its range is fetched (mi355lz4_decompress_linked_end_last);
    "segments": the pointer pass is given 8 blocks at a time, so a range is several segments, the last block is not
    available ahead and the call must say so (the driver then finishes the range first); "runin": the first range through
    the run-in decode."""
    env = {"MI355LZ4_LINKED_PTR_BLOCKS": "8"} if early == "segments" else None
    if early == "runin":
        # the stream's first range has no seam to wait for: it is finished in _begin by the run-in decode (forced here: by
        # default only ranges of 9 216 blocks and more), pieces of 3 with a run-in of 2 blocks; the other range as always
        env = {"MI355LZ4_LINKED_RUNIN": "1", "MI355LZ4_LINKED_RUNIN_PIECE": "3", "MI355LZ4_LINKED_RUNIN_BLOCKS": "2", "MI355LZ4_LINKED_RUNS": "0"}
    early = bool(early)
    world = len(cuts) - 1
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29710 + n_blocks + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, kind, n_blocks, cuts, q, early, env)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = dict(q.get(timeout=5) for _ in range(world))
    assert got == {r: True for r in range(world)}


def test_begin_end_equals_one_call(engine, oracle):
    """The two halves of the C ABI, back to back, against the one-call linked decode (and the input)."""
    raw, framed, offs = _stream(oracle, "text", 64, 9)
    n = 64
    fr = torch.from_numpy(np.frombuffer(framed, dtype=np.uint8).copy()).cuda()
    boff = torch.tensor(offs, dtype=torch.int64).cuda()
    ooff = (torch.arange(n + 1, dtype=torch.int64) * BL).cuda()
    out1 = torch.zeros(n * BL, dtype=torch.uint8, device="cuda")
    res1 = torch.zeros(n, dtype=torch.int32, device="cuda")
    engine.decompress_batch_device(fr, len(framed), boff, n, out1, ooff, res1, linked=True)
    out2 = torch.zeros(n * BL, dtype=torch.uint8, device="cuda")
    res2 = torch.zeros(n, dtype=torch.int32, device="cuda")
    engine.decompress_linked_begin(fr, len(framed), boff, n, out2, ooff, res2, 0)
    engine.decompress_linked_end()
    engine.synchronize()
    assert res1.cpu().tolist() == [BL] * n and torch.equal(res1, res2)
    assert out1.cpu().numpy().tobytes() == raw and torch.equal(out1, out2)
    # a range in the middle of the stream with its seam placed between the calls
    b0 = 17
    sub = framed[offs[b0]:]
    fr3 = torch.from_numpy(np.frombuffer(sub, dtype=np.uint8).copy()).cuda()
    boff3 = torch.tensor([o - offs[b0] for o in offs[b0:]], dtype=torch.int64).cuda()
    m = n - b0
    ooff3 = (torch.arange(m + 2, dtype=torch.int64) * BL).cuda()            # [seam][blocks...]
    out3 = torch.zeros((m + 1) * BL, dtype=torch.uint8, device="cuda")
    res3 = torch.zeros(m + 1, dtype=torch.int32, device="cuda")
    res3[0] = BL
    engine.decompress_linked_begin(fr3, len(sub), boff3, m, out3, ooff3, res3, 1)
    out3[:BL].copy_(out1[(b0 - 1) * BL: b0 * BL])                           # the seam arrives
    # the range's last block ahead of the others: final before _end has run
    assert engine.decompress_linked_end_last() is True
    assert out3[m * BL:].cpu().numpy().tobytes() == raw[(n - 1) * BL:]
    assert not torch.equal(out3[BL:], out1[b0 * BL:])                       # (the rest of the range is not there yet)
    engine.decompress_linked_end()
    engine.synchronize()
    assert res3.cpu().tolist() == [BL] * (m + 1)
    assert out3[BL:].cpu().numpy().tobytes() == raw[b0 * BL:]
    # nothing begun, or a range in which no block needs its dictionary: the last block is final as it is
    assert engine.decompress_linked_end_last() is True
