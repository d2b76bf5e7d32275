"""Ordered multi-GPU gather of ragged per-rank block streams.

The framed block stream shards by per-block round-robin: global block k lives on
rank k % G as that rank's local block k // G.  Compute needs no collective.  To
hand one in-order stream to a single consumer:

  1. all_gather of the per-block byte counts (int32)  -> every rank knows every size;
  2. each peer sends its dense local buffer to the root over its direct xGMI link
     (torch.distributed isend/irecv == ncclSend/ncclRecv on the RCCL backend; no ring);
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


def gather_ordered(local, local_sizes, root=0, engine=None, group=None):
    """Gather every rank's dense local stream (uint8 tensor `local`, per-block byte counts
    `local_sizes` int32) into one in-order stream on `root`.  Returns (stream, global_off) on the
    root and (None, global_off) elsewhere."""
    rank = dist.get_rank(group)
    G = dist.get_world_size(group)
    dev = local.device
    sizes, goff = global_layout(local_sizes, group)
    loff = torch.zeros(local_sizes.numel() + 1, dtype=torch.int64, device=dev)
    torch.cumsum(local_sizes.to(torch.int64), 0, out=loff[1:])
    if rank != root:
        n = int(loff[-1].item())
        dist.isend(local[:n].contiguous(), dst=root, group=group).wait()
        return None, goff
    total = int(goff[-1].item())
    out = torch.empty(total, dtype=torch.uint8, device=dev)
    pending = []
    for g in range(G):
        if g == root:
            continue
        n = int(sizes[g].to(torch.int64).sum().item())
        stage = torch.empty(n, dtype=torch.uint8, device=dev)
        pending.append((g, stage, dist.irecv(stage, src=g, group=group)))
    # The engine launches on torch's current stream (Engine._follow_torch), the stream req.wait() orders the
    # receive against: no host synchronisation is needed between a receive and its interleave.
    interleave(local, loff, root, G, out, goff, engine)                  # overlaps with the receives
    for g, stage, req in pending:                                         # each peer is placed as soon as it has arrived
        req.wait()
        poff = torch.zeros(sizes[g].numel() + 1, dtype=torch.int64, device=dev)
        torch.cumsum(sizes[g].to(torch.int64), 0, out=poff[1:])
        interleave(stage, poff, g, G, out, goff, engine)
    del pending
    if out.is_cuda:
        torch.cuda.current_stream().synchronize()
    return out, goff
