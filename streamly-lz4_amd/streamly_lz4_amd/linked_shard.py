"""ONE linked stream decoded by several GPUs (DESIGN.md "Multi-GPU").

A stream the reference's compressor wrote does not shard by round-robin: block k's dictionary is the output of
block k-1 (reference cbits/lz4.c:2347-2355, kept alive by src/Streamly/Internal/LZ4.hs:564).  It shards by
CONTIGUOUS RANGES of blocks, one per rank, and the only thing a rank needs from its left neighbour is the output of
the block in front of its range -- the seam, one block.

Every rank issues, at once and without waiting for anybody, the part of the linked decode that reads no output
byte of the seam: the standalone pass, the tolerant re-decode, source pointers and pointer jumping over its whole
range (mi355lz4_decompress_linked_begin).  Then, rank by rank, the seam arrives, the bytes are fetched from their
roots (mi355lz4_decompress_linked_end) and the rank's last block goes to its right neighbour: the serial part of a
stream over G GPUs is G fetch passes and G - 1 messages of one block, not G ranges.

The exchange is a point-to-point send of <= 64 KiB per rank (torch.distributed: ncclSend/ncclRecv on the RCCL
backend; staged through the host on gloo, which the tests use to rehearse two ranks on one GPU).
"""
import torch
import torch.distributed as dist


def decode_linked_sharded(engine, framed, framed_len, block_off, ulen, group=None, header_kind=8, fixed_uncomp=0,
                          use_end_last=True):
    """This rank's contiguous range of ONE linked stream.

    framed     uint8 device tensor: the range's framed blocks, dense
    block_off  int64 device tensor (n + 1): offsets of the block headers in `framed`
    ulen       int64 host tensor / list (n): decoded size of each block (the header's uncompLen, or the capacity)
    use_end_last  hand the range's last block on before the rest of the range is fetched (False: after, for the tests)
    Returns (out, result): the range's decoded bytes (blocks back to back) and the per-block results (int32)."""
    rank = dist.get_rank(group)
    G = dist.get_world_size(group)
    dev = framed.device
    ulen = torch.as_tensor(ulen, dtype=torch.int64)
    n = int(ulen.numel())
    on_host = dist.get_backend(group) == "gloo"
    # every rank learns the size of the block in front of its range
    lastLen = torch.tensor([int(ulen[-1]) if n else 0], dtype=torch.int64, device="cpu" if on_host else dev)
    lens = [torch.zeros_like(lastLen) for _ in range(G)]
    dist.all_gather(lens, lastLen, group=group)
    seam = int(lens[rank - 1].item()) if rank > 0 else 0
    lb = 1 if rank > 0 else 0
    if rank > 0 and seam <= 0:
        raise ValueError("decode_linked_sharded: a range must not follow an empty block")
    # output: [seam][block 0][block 1]...; out_off / result carry the leading entry for the seam
    sizes = torch.cat([torch.tensor([seam], dtype=torch.int64), ulen])
    out_off = torch.zeros(n + 2, dtype=torch.int64)
    torch.cumsum(sizes, 0, out=out_off[1:])
    total = int(out_off[-1].item())
    out = torch.empty(max(total, 1), dtype=torch.uint8, device=dev)
    out_off_d = out_off.to(dev)
    result = torch.zeros(n + 1, dtype=torch.int32, device=dev)
    result[0] = seam
    req = stage = None
    if rank > 0:
        stage = torch.empty(seam, dtype=torch.uint8, device="cpu" if on_host else dev)
        req = dist.irecv(stage, src=rank - 1, group=group)
    if lb:
        engine.decompress_linked_begin(framed, framed_len, block_off, n, out, out_off_d, result, 1,
                                       header_kind=header_kind, fixed_uncomp=fixed_uncomp)
    else:
        engine.decompress_linked_begin(framed, framed_len, block_off, n, out, out_off_d[1:].contiguous(),
                                       result[1:], 0, header_kind=header_kind, fixed_uncomp=fixed_uncomp)
    if rank > 0:
        req.wait()
        out[:seam].copy_(stage, non_blocking=False)
        if getattr(engine, "_pinned", False):
            torch.cuda.current_stream(dev).synchronize()
    def send_last():
        if getattr(engine, "_pinned", False):
            engine.synchronize()
        last = out[int(out_off[n].item()): int(out_off[n + 1].item())]
        if on_host:
            engine.synchronize()
            dist.send(last.cpu(), dst=rank + 1, group=group)
        else:
            dist.send(last.contiguous(), dst=rank + 1, group=group)

    # The right neighbour waits for ONE block: the range's last block is fetched ahead of the others and sent on, the
    # rest of the range is fetched while the neighbours further right take their turn.  (When that block is not
    # available ahead -- a chain deeper than the chasing fetch follows, a range in several segments -- the whole range
    # is finished first, as before.)
    early = rank + 1 < G and n > 0 and use_end_last and engine.decompress_linked_end_last()
    if early:
        send_last()
    engine.decompress_linked_end()
    if rank + 1 < G and n > 0 and not early:
        send_last()
    engine.synchronize()
    return out[seam:total], result[1:]
