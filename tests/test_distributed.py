"""CPU, world_size 2 over gloo: the N>1 path's ordered gather (round-robin shards -> one in-order
stream on the root).  The codec itself needs no collective; this is the only exchange step."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, n_global, seed, q, stage_bytes=None):
    for p in (ROOT, os.path.join(ROOT, "streamly-lz4_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from streamly_lz4_amd.gather import gather_ordered
        rng = np.random.default_rng(seed)
        sizes = rng.integers(1, 200, size=n_global)
        blocks = [rng.integers(0, 256, size=int(s), dtype=np.uint8) for s in sizes]   # same on every rank
        mine = [blocks[k] for k in range(rank, n_global, world)]                       # block k -> rank k % G
        local = torch.from_numpy(np.concatenate(mine))
        lsz = torch.tensor([len(b) for b in mine], dtype=torch.int32)
        out, goff = gather_ordered(local, lsz, root=0, stage_bytes=stage_bytes)
        if rank == 0:
            want = np.concatenate(blocks)
            ok = out is not None and np.array_equal(out.numpy(), want) and \
                goff.tolist() == [0] + np.cumsum(sizes).tolist()
            q.put(bool(ok))
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_global,stage_bytes", [(2, None), (10, None), (64, None), (64, 300), (66, 1), (200, 1000)])
def test_gather_ordered_world2(n_global, stage_bytes):
    """stage_bytes: the peer's stream crosses in pieces of whole blocks (two staging buffers on the root); 1 = a
    piece per block."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + n_global + (stage_bytes or 0) % 97
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_global, 7, q, stage_bytes)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


@pytest.mark.parametrize("n_global,stage_bytes", [(9, None), (63, 300), (66, 1)])
def test_gather_ordered_world3(n_global, stage_bytes):
    """Two peers per piece round (one batch of receives on the root), with unequal piece counts per peer."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29800 + n_global + (stage_bytes or 0) % 97
    procs = [ctx.Process(target=_worker, args=(r, 3, port, n_global, 11, q, stage_bytes)) for r in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
