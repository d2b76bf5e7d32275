"""Ordered multi-GPU gather of ragged per-rank block streams.

The framed block stream shards by per-block round-robin: global block k lives on
rank k % G as that rank's local block k // G.  Compute needs no collective.  To
hand one in-order stream to a single consumer:

  1. all_gather of the per-block byte counts (int32)  -> every rank knows every size;
  2. each peer sends its dense local buffer to the root over its direct xGMI link, every peer's transfer of a round
     in ONE batch (dist.batch_isend_irecv == ncclGroupStart / ncclSend, ncclRecv / ncclGroupEnd on RCCL; no ring);
  3. the root scatters block (g, j) to globalOff[j*G + g] -- the HIP interleave
     kernel (mi355lz4_interleave_device) on the GPU.

On CPU tensors (gloo; used by the world_size-2 tests) step 3 is plain tensor
slicing: it is data placement, not codec work.
"""
import torch
import torch.distributed as dist


def global_layout(local_sizes, group=None):
    """all_gather the per-block sizes.  Returns (sizes[G][n_local], global_off[n_global+1]) on every rank;
    all ranks must hold the same number of local blocks."""
    G = dist.get_world_size(group)
    sizes = [torch.empty_like(local_sizes) for _ in range(G)]
    dist.all_gather(sizes, local_sizes.contiguous(), group=group)
    stacked = torch.stack(sizes, dim=1).reshape(-1).to(torch.int64)      # order j*G + g == global block index
    off = torch.zeros(stacked.numel() + 1, dtype=torch.int64, device=local_sizes.device)
    torch.cumsum(stacked, 0, out=off[1:])
    return sizes, off


def interleave(local, local_off, rank, n_ranks, global_buf, global_off, engine=None):
    """Place rank-local dense blocks into the global stream (root side)."""
    n_local = local_off.numel() - 1
    if local.is_cuda:
        if engine is None:
            raise RuntimeError("interleave on device tensors needs the HIP engine (no fallback)")
        engine.interleave_device(local, local_off, n_local, rank, n_ranks, global_buf, global_off)
        return
    lo = local_off.tolist()
    go = global_off.tolist()
    for j in range(n_local):
        n = lo[j + 1] - lo[j]
        d = go[j * n_ranks + rank]
        global_buf[d:d + n] = local[lo[j]:lo[j + 1]]


STAGE_BYTES = 256 << 20        # a peer's stream crosses the link in pieces of at most this many bytes (whole blocks)


def _pieces(sizes64, limit):
    """Cut blocks 0..n into runs of consecutive blocks of at most `limit` bytes (at least one block each).
    Every rank computes the same cut from the all-gathered sizes.  Returns [(j0, j1, byte0, byte1)]."""
    off = [0]
    for v in sizes64.tolist():
        off.append(off[-1] + int(v))
    out, j0 = [], 0
    n = len(off) - 1
    while j0 < n:
        j1 = j0 + 1
        while j1 < n and off[j1 + 1] - off[j0] <= limit:
            j1 += 1
        out.append((j0, j1, off[j0], off[j1]))
        j0 = j1
    return out


def _order_engine_after_torch(engine, tensor):
    """The interleave kernel runs on the ENGINE's stream; the receive it reads was ordered against torch's current
    stream by req.wait().  They are the same stream while the engine follows torch (the default); an engine pinned
    to a stream of its own (Engine.use_stream) has to wait for torch's stream first."""
    if engine is not None and tensor.is_cuda and getattr(engine, "_pinned", False):
        torch.cuda.current_stream(tensor.device).synchronize()


def gather_ordered(local, local_sizes, root=0, engine=None, group=None, stage_bytes=None):
    """Gather every rank's dense local stream (uint8 tensor `local`, per-block byte counts
    `local_sizes` int32) into one in-order stream on `root`.  Returns (stream, global_off) on the
    root and (None, global_off) elsewhere.

    A peer's stream crosses its link in pieces of whole blocks (<= stage_bytes, default 256 MiB); the root keeps two
    staging buffers per peer and places a piece (HIP interleave kernel) while the next one is in flight, so its
    staging memory is 2 x stage_bytes per peer whatever the streams' sizes (round 2 staged every peer in full)."""
    rank = dist.get_rank(group)
    G = dist.get_world_size(group)
    dev = local.device
    limit = int(stage_bytes or STAGE_BYTES)
    sizes, goff = global_layout(local_sizes, group)
    sizes64 = [x.to(torch.int64).cpu() for x in sizes]
    # The transfers of one piece ROUND -- piece k of every peer that has one -- are posted as ONE batch
    # (dist.batch_isend_irecv == ncclGroupStart ... ncclGroupEnd on the RCCL backend): every peer's
    # send/receive pair of the round is in flight at once, each on the peer's own xGMI link.  Posted one by one, point
    # to point operations on one communicator run in posting order and the root would drain peer 1 before peer 2.
    # (The piece number is also the message tag: gloo does not match equal tags in order; RCCL ignores tags and is FIFO.)
    def batch(ops):
        return dist.batch_isend_irecv(ops) if ops else []

    if rank != root:
        pcs = [pc for pc in _pieces(sizes64[rank], limit) if pc[3] > pc[2]]
        reqs = []
        for k, (_j0, _j1, b0, b1) in enumerate(pcs):           # a peer queues all its rounds without waiting
            reqs += batch([dist.P2POp(dist.isend, local[b0:b1].contiguous(), root, group=group, tag=k)])
        for r in reqs:
            r.wait()
        return None, goff
    total = int(goff[-1].item())
    out = torch.empty(total, dtype=torch.uint8, device=dev)
    loff = torch.zeros(local_sizes.numel() + 1, dtype=torch.int64, device=dev)
    torch.cumsum(local_sizes.to(torch.int64), 0, out=loff[1:])
    peers = []
    for g in range(G):
        if g == root:
            continue
        poff = torch.zeros(sizes[g].numel() + 1, dtype=torch.int64, device=dev)
        torch.cumsum(sizes[g].to(torch.int64), 0, out=poff[1:])
        pcs = [pc for pc in _pieces(sizes64[g], limit) if pc[3] > pc[2]]
        cap = max([pc[3] - pc[2] for pc in pcs], default=0)
        peers.append({"g": g, "poff": poff, "pieces": pcs,
                      "stages": [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(min(2, len(pcs)))]})
    rounds = max([len(p["pieces"]) for p in peers], default=0)

    def post_round(k):
        ops, meta = [], []
        for p in peers:
            if k < len(p["pieces"]):
                _j0, _j1, b0, b1 = p["pieces"][k]
                stage = p["stages"][k % len(p["stages"])][: b1 - b0]
                ops.append(dist.P2POp(dist.irecv, stage, p["g"], group=group, tag=k))
                meta.append((p, stage))
        return k, meta, batch(ops)

    inflight = [post_round(k) for k in range(min(2, rounds))]    # two staging buffers per peer: two rounds in flight
    interleave(local, loff, root, G, out, goff, engine)          # overlaps with the receives
    while inflight:
        k, meta, reqs = inflight.pop(0)
        for r in reqs:
            r.wait()
        for p, stage in meta:
            _order_engine_after_torch(engine, stage)
            j0, j1, b0, b1 = p["pieces"][k]
            interleave(stage, p["poff"][j0:j1 + 1] - b0, p["g"], G, out, goff[j0 * G:], engine)
        if k + 2 < rounds:
            # round k + 2 receives into the staging buffers round k used: the kernels that read them must be done first
            if out.is_cuda and engine is not None:
                engine.synchronize()
            inflight.append(post_round(k + 2))
    if out.is_cuda:
        if engine is not None:
            engine.synchronize()
        torch.cuda.current_stream(dev).synchronize()
    return out, goff
