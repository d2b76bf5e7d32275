#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X LZ4 block engine.

    python bench.py --gpus N --steps K --warmup W [--workload decompress|compress|roundtrip|random256k|text]

One "step" = one pass of the hot path over one batch of input resident in HBM.  Default workload (N=1)
is BASELINE.json configs[1]: decompress-only, 64 KiB blocks, 4 GiB lzsynth(16, 2048) stream of
independent blocks (compressed by this engine during setup).  For N>1 (one process per GPU under
torch.distributed.run, RCCL) block k of the global stream lives on rank k % N (per-block round-robin);
every rank processes 4 GiB (weak scaling) with no data-path collective.  Rank 0 prints ONE JSON line.

`value` is whole-job GB/s of UNCOMPRESSED bytes over the wall-clock of the K timed steps (max over ranks).
`roofline` prices the dominant kernel of the timed phase: algorithmic bytes (U + C) per
launch / its average launch duration, measured live with HIP events on the stream the kernels are launched
on, against the 8 TB/s HBM peak.  The metric's name says compress+decompress, so the same line also carries
`roundtrip` (compress + compact + decompress of the same data, timed after the K steps) and `compress`
(the encoder's own roofline block).  `cpu_baseline` times the reference codec (oracle/_ref, kind
"reference"; or the oracle port) on ONE host core on a bounded sample of the same input -- context, not
the target; its `ratio` is the reference's compressed size for the same data.

N>1 also runs the RCCL ordered gather of the framed output once (untimed by `value`), decodes the gathered
stream on the root and compares it with the generator's global stream, and reports compute-only and
compute+gather rates, the gather's own rate against the root's xGMI links and `roundtrip_plus_gather_GBps`.

BASELINE config 4 (64 GiB over 8 GPUs) is 8 GiB per GPU: `--workload roundtrip --blocks 131072` runs that share
(131 072 blocks of 64 KiB compressed, compacted and decoded in one call each); rehearsed at N = 1
(profiles/r05_bench_roundtrip_8GiB.json).  At N = 8 the root's verification decodes 8 x that in one call.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    # name: (kind, block_len, n_blocks_per_gpu, accel, timed phase)
    "decompress": ("lzsynth", 65536, 65536, 1, "decompress"),   # BASELINE configs[1]
    "compress": ("lzsynth", 65536, 65536, 1, "compress"),       # configs[2] shape; Canterbury "large" when present
    "roundtrip": ("lzsynth", 65536, 65536, 1, "roundtrip"),     # configs[3] per-GPU share
    "random256k": ("random", 262144, 16384, 400, "roundtrip"),  # configs[4] per-GPU share
    "text": ("text", 65536, 65536, 1, "decompress"),
}


def canterbury_large(n_bytes):
    """configs[2] input: bible.txt, E.coli, world192.txt of the Canterbury "large" corpus, each cycled like
    the reference's benchmark cycles its files (benchmark/Main.hs:80-84), laid end to end.  None if absent."""
    try:
        import corpus
    except Exception:
        return None, None
    paths = [corpus.find(r) for r in ("large/bible.txt", "large/E.coli", "large/world192.txt")]
    if not all(paths):
        return None, None
    share = n_bytes // len(paths) // 65536 * 65536
    parts = [corpus.cycled(p, share) for p in paths[:-1]]
    parts.append(corpus.cycled(paths[-1], n_bytes - share * (len(paths) - 1)))
    return b"".join(parts), "Canterbury large (bible.txt, E.coli, world192.txt cycled)"


def reference_stream_decode(S, eng, torch, dev, comps, src, ns, BL, linked=True, copies=1):
    """GPU decode of a stream of `ns` blocks written by the reference's compressor: its own linked stream (linked = 1,
    the only thing compressChunks ever writes) or the same blocks compressed independently (linked = 0).  copies > 1:
    the stream `copies` times over as ONE stream of copies * ns blocks -- valid because the reference wrote the sample's
    first block without a dictionary -- to show the rate on a stream longer than the CPU-timed sample."""
    import struct
    import numpy as np
    framed = b"".join(struct.pack("<ii", len(c), BL) + c for c in comps)
    offs1 = np.zeros(ns + 1, dtype=np.int64)
    np.cumsum([8 + len(c) for c in comps], out=offs1[1:])
    nt = ns * copies
    offs = np.concatenate([offs1[:-1] + k * len(framed) for k in range(copies)] + [np.array([copies * len(framed)], dtype=np.int64)])
    one = torch.from_numpy(np.frombuffer(framed, dtype=np.uint8).copy()).to(dev)
    buf = one.repeat(copies) if copies > 1 else one
    off = torch.from_numpy(offs).to(dev)
    ooff = torch.arange(nt + 1, dtype=torch.int64, device=dev) * BL
    out = torch.zeros(nt * BL, dtype=torch.uint8, device=dev)
    res = torch.zeros(nt, dtype=torch.int32, device=dev)
    eng.decompress_batch_device(buf, len(framed) * copies, off, nt, out, ooff, res, linked=False)
    eng.synchronize()
    dependent = int((res < 0).sum().item())
    e0, e1 = S.Event(), S.Event()
    best = 1e9
    for _ in range(3):
        eng.record(e0)
        eng.decompress_batch_device(buf, len(framed) * copies, off, nt, out, ooff, res, linked=linked)
        eng.record(e1)
        eng.synchronize()
        best = min(best, eng.elapsed_ms(e0, e1))
    ok = bool((res == BL).all().item()) and all(torch.equal(out[k * ns * BL:(k + 1) * ns * BL], src[: ns * BL]) for k in range(copies))
    if not ok:
        sys.exit("bench.py: the reference-written %s stream does not decode to the input" % ("linked" if linked else "independent"))
    r = {"blocks": nt, "dependent_blocks": dependent, "ms": round(best, 3), "GBps": round(nt * BL / best / 1e6, 2),
         "verified": True}
    if linked:
        r["note"] = "one linked stream, one call; device-resident, HIP events" + (
            "; the CPU-timed sample's stream %d times over (each copy starts with the one block the reference wrote without a dictionary)" % copies
            if copies > 1 else "")
    else:
        r["ratio"] = round(ns * BL / len(framed), 4)
        r["frac"] = round((ns * BL + len(framed)) / best / 1e6 / HBM_PEAK_GBPS, 5)
        r["note"] = ("the same blocks compressed INDEPENDENTLY by the reference codec on the host (fresh context per block), "
                     "decoded by the kernel `value` times; device-resident, HIP events, best of 3")
    return r


def incompressible_rates(S, eng, torch, dev):
    """The north star's "compressible and incompressible": 4 GiB of random bytes in 256 KiB blocks at acceleration 400
    (BASELINE configs[4]'s per-GPU share), compress and decompress timed with HIP events, round trip checked."""
    BL, NB, accel = 262144, 16384, 400
    U = NB * BL
    src = torch.empty(U, dtype=torch.uint8, device=dev)
    eng.generate("random", src, BL, NB)
    stride = S.slot_stride(BL, 8)
    slots = torch.empty(NB * stride, dtype=torch.uint8, device=dev)
    flen = torch.empty(NB, dtype=torch.int32, device=dev)
    doff = torch.empty(NB + 1, dtype=torch.int64, device=dev)
    dense = torch.empty(NB * stride, dtype=torch.uint8, device=dev)
    ooff = torch.arange(NB + 1, dtype=torch.int64, device=dev) * BL
    out = torch.empty(U, dtype=torch.uint8, device=dev)
    res = torch.empty(NB, dtype=torch.int32, device=dev)
    ev = [S.Event() for _ in range(3)]
    tc = td = 1e9
    C = 0
    for _ in range(3):
        eng.record(ev[0])
        eng.compress_batch_device(src, NB, BL, slots, stride, flen, accel=accel)
        eng.record(ev[1])
        eng.compact_device(slots, stride, flen, NB, dense, NB * stride, doff)
        eng.synchronize()
        C = int(doff[-1].item())
        tc = min(tc, eng.elapsed_ms(ev[0], ev[1]))
        eng.record(ev[1])
        eng.decompress_batch_device(dense, C, doff, NB, out, ooff, res)
        eng.record(ev[2])
        eng.synchronize()
        td = min(td, eng.elapsed_ms(ev[1], ev[2]))
    if not (bool((res == BL).all().item()) and torch.equal(out, src)):
        sys.exit("bench.py: incompressible round trip mismatch")
    return {"workload": "random bytes, 256 KiB blocks, %d blocks (4 GiB), accel %d" % (NB, accel), "ratio": round(U / C, 5),
            "decompress_GBps": round(U / td / 1e6, 2), "decompress_frac": round((U + C) / td / 1e6 / HBM_PEAK_GBPS, 5),
            "compress_GBps": round(U / tc / 1e6, 2), "compress_frac": round((U + C) / tc / 1e6 / HBM_PEAK_GBPS, 5),
            "verified": True, "note": "HIP events, best of 3, device-resident"}


def small_call_rates(S, eng, torch, dev, src, BL, kind, accel, with_cpu):
    """The reference's own benchmark protocol is a SMALL call: 10 MiB per file (benchmark/Main.hs:80-84) = 160 blocks of
    64 KiB.  The first 10 MiB of the bench's stream through one call each way: device-resident (HIP events; the decoder the
    engine picks for a call of this size -- one workgroup per block, csrc/decode_cu.hpp -- and, beside it, the
    one-wavefront-per-block decoder the 4 GiB figure is measured with), host buffer to host buffer through the C ABI
    (pageable memory, wall clock), and the reference codec on the SAME bytes on one host core and on all of them."""
    import ctypes as C
    import numpy as np
    NB = min((10 << 20) // BL, src.numel() // BL)
    U = NB * BL
    stride = S.slot_stride(BL, 8)
    slots = torch.empty(NB * stride, dtype=torch.uint8, device=dev)
    flen = torch.empty(NB, dtype=torch.int32, device=dev)
    doff = torch.empty(NB + 1, dtype=torch.int64, device=dev)
    dense = torch.empty(NB * stride, dtype=torch.uint8, device=dev)
    ooff = torch.arange(NB + 1, dtype=torch.int64, device=dev) * BL
    out = torch.empty(U, dtype=torch.uint8, device=dev)
    res = torch.empty(NB, dtype=torch.int32, device=dev)
    ev = [S.Event() for _ in range(3)]
    tc = 1e9
    Cb = 0
    for _ in range(5):
        eng.record(ev[0])
        eng.compress_batch_device(src, NB, BL, slots, stride, flen, accel=accel)
        eng.compact_device(slots, stride, flen, NB, dense, NB * stride, doff)
        eng.record(ev[1])
        eng.synchronize()
        Cb = int(doff[-1].item())
        tc = min(tc, eng.elapsed_ms(ev[0], ev[1]))
    td = {}
    for name, variant in (("decompress_ms", 0), ("decompress_one_wavefront_per_block_ms", 2)):
        eng.set_decoder(variant)
        best = 1e9
        for _ in range(8):
            out.zero_()
            eng.record(ev[1])
            eng.decompress_batch_device(dense, Cb, doff, NB, out, ooff, res)
            eng.record(ev[2])
            eng.synchronize()
            best = min(best, eng.elapsed_ms(ev[1], ev[2]))
        if not (bool((res == BL).all().item()) and torch.equal(out, src[:U])):
            sys.exit("bench.py: small-call round trip mismatch (decoder %d)" % variant)
        td[name] = round(best, 4)
    eng.set_decoder(0)
    r = {"workload": "first %d blocks of %d KiB (%.1f MiB) of the same %s stream, one call each way" % (NB, BL >> 10, U / 2 ** 20, kind),
         "ratio": round(U / Cb, 4),
         "device_resident": dict({"compress_ms": round(tc, 4)}, **td),
         "note": "reference protocol: benchmark/Main.hs:80-84 (10 MiB per file); best of 5 / 8 calls"}
    # host buffer to host buffer through the C ABI (what the Haskell shim's one FFI call per batch costs)
    L = S.lib
    u8p, i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
    cap = NB * (S.compress_bound(BL) + 8)
    for memory in ("pageable", "pinned"):
        # (pinned: page-locked caller buffers, what a binding that allocates its arrays through the engine gets -- no staging copies)
        host_t = torch.empty(U, dtype=torch.uint8); framed_t = torch.empty(cap, dtype=torch.uint8); out_t = torch.empty(U, dtype=torch.uint8)
        if memory == "pinned":
            host_t, framed_t, out_t = host_t.pin_memory(), framed_t.pin_memory(), out_t.pin_memory()
        host_t.copy_(src[:U].cpu())
        ptrs = (u8p * NB)(*[C.cast(host_t.data_ptr() + i * BL, u8p) for i in range(NB)])
        lens = np.full(NB, BL, dtype=np.int32)
        fl = np.zeros(NB, dtype=np.int32); st = np.zeros(NB, dtype=np.int32); bl = np.zeros(NB, dtype=np.int32)
        olen, dlen, got = C.c_size_t(), C.c_size_t(), C.c_int()
        hc = hd = 1e9
        key = "host_to_host" if memory == "pageable" else "host_to_host_pinned"
        for _ in range(7):
            t0 = time.perf_counter()
            rc = L.mi355lz4_compress_batch(eng.ctx, ptrs, lens.ctypes.data_as(i32p), NB, accel, 8, C.cast(framed_t.data_ptr(), u8p), cap,
                                           C.byref(olen), fl.ctypes.data_as(i32p), st.ctypes.data_as(i32p))
            t1 = time.perf_counter()
            rc2 = L.mi355lz4_decompress_batch(eng.ctx, C.cast(framed_t.data_ptr(), u8p), olen.value, 8, 0, 0, None, 0,
                                              C.cast(out_t.data_ptr(), u8p), U, C.byref(dlen), bl.ctypes.data_as(i32p), NB, C.byref(got))
            t2 = time.perf_counter()
            if rc != 0 or rc2 != 0 or dlen.value != U:
                r[key] = {"error": (L.mi355lz4_last_error() or b"").decode()}
                return r
            hc, hd = min(hc, t1 - t0), min(hd, t2 - t1)
        if not torch.equal(out_t, host_t):
            sys.exit("bench.py: small-call host round trip mismatch")
        r[key] = {"compress_ms": round(hc * 1e3, 4), "decompress_ms": round(hd * 1e3, 4), "memory": memory}
    if with_cpu:
        from oracle import oracle as orc
        hostb = host_t.numpy()
        blocks = [hostb[i * BL:(i + 1) * BL].tobytes() for i in range(NB)]
        one = orc.cpu_baseline(blocks, accel=accel)
        allc = orc.cpu_baseline_all_cores(blocks, accel=accel)
        r["cpu_same_bytes"] = {"kind": one["kind"],
                               "one_core": {"compress_ms": round(one["comp_s"] * 1e3, 4), "decompress_ms": round(one["decomp_s"] * 1e3, 4)},
                               "all_cores": {"cores": allc["threads"], "compress_ms": round(allc["comp_s"] * 1e3, 4),
                                             "decompress_ms": round(allc["decomp_s"] * 1e3, 4),
                                             "note": "best-case CPU, not reference behaviour: one stream per thread"}}
    return r


def host_api_rates(S, eng, src, BL, kind):
    """PCIe-inclusive rate of the host-buffer C API (what the Haskell shim binds) on the first 512 MiB of the
    same stream: pageable caller memory (staged through pinned slots) and page-locked caller memory (DMA
    straight from / to the caller's buffers).  Never `value`: the inputs start in host memory here."""
    import ctypes as C
    import numpy as np
    import torch
    L = S.lib
    u8p, i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
    nb = min(src.numel() // BL, (512 << 20) // BL)
    cap = nb * (S.compress_bound(BL) + 8)
    res = {"MiB_per_call": nb * BL >> 20, "link_GBps_one_direction": None}
    try:
        h = torch.empty(256 << 20, dtype=torch.uint8).pin_memory()
        d = torch.empty(256 << 20, dtype=torch.uint8, device=src.device)
        d.copy_(h, non_blocking=True); torch.cuda.synchronize()
        t0 = time.perf_counter(); d.copy_(h, non_blocking=True); torch.cuda.synchronize()
        res["link_GBps_one_direction"] = round(h.numel() / (time.perf_counter() - t0) / 1e9, 1)
        del h, d
    except Exception:
        pass
    for mem in ("pageable", "pinned"):
        mk = (lambda n: torch.empty(n, dtype=torch.uint8).pin_memory()) if mem == "pinned" else (lambda n: torch.empty(n, dtype=torch.uint8))
        host_t, framed_t, out_t = mk(nb * BL), mk(cap), mk(nb * BL)
        host_t.copy_(src[: nb * BL].cpu())
        ptrs = (u8p * nb)(*[C.cast(host_t.data_ptr() + i * BL, u8p) for i in range(nb)])
        lens = np.full(nb, BL, dtype=np.int32)
        flen = np.zeros(nb, dtype=np.int32); st = np.zeros(nb, dtype=np.int32); blen = np.zeros(nb, dtype=np.int32)
        olen, dlen, got = C.c_size_t(), C.c_size_t(), C.c_int()
        tc = td = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            rc = L.mi355lz4_compress_batch(eng.ctx, ptrs, lens.ctypes.data_as(i32p), nb, 1, 8, C.cast(framed_t.data_ptr(), u8p), cap,
                                           C.byref(olen), flen.ctypes.data_as(i32p), st.ctypes.data_as(i32p))
            t1 = time.perf_counter()
            if rc != 0:
                return {"error": (L.mi355lz4_last_error() or b"").decode()}
            rc = L.mi355lz4_decompress_batch(eng.ctx, C.cast(framed_t.data_ptr(), u8p), olen.value, 8, 0, 1, None, 0,
                                             C.cast(out_t.data_ptr(), u8p), nb * BL, C.byref(dlen), blen.ctypes.data_as(i32p), nb, C.byref(got))
            t2 = time.perf_counter()
            if rc != 0 or dlen.value != nb * BL:
                return {"error": (L.mi355lz4_last_error() or b"").decode()}
            tc, td = min(tc, t1 - t0), min(td, t2 - t1)
        if not torch.equal(out_t, host_t):
            return {"error": "host API round trip mismatch"}
        res[mem] = {"compress_GBps": round(nb * BL / tc / 1e9, 2), "decompress_GBps": round(nb * BL / td / 1e9, 2)}
    return res


def one_stream_bench(args, torch, S, dist, eng, dev, rank, world, red_dev, kind, BL, NB, accel):
    """ONE linked stream over all ranks: rank r holds blocks [r NB, (r + 1) NB) of the generator's
    stream, compressed as one linked stream (block k's dictionary is block k - 1, across the ranks' seams too), and
    decodes its range with linked_shard.decode_linked_sharded.  Prints one JSON line (rank 0)."""
    from streamly_lz4_amd.linked_shard import decode_linked_sharded
    lb = 1 if rank > 0 else 0                           # the block in front of my range: my first block's dictionary
    n = NB + lb
    src = torch.empty(n * BL, dtype=torch.uint8, device=dev)
    eng.generate(kind, src, BL, n, first_block=rank * NB - lb, block_step=1)
    stride = S.slot_stride(BL, 8)
    slots = torch.empty(n * stride, dtype=torch.uint8, device=dev)
    flen = torch.empty(n, dtype=torch.int32, device=dev)
    doff = torch.empty(n + 1, dtype=torch.int64, device=dev)
    dense = torch.empty(n * stride, dtype=torch.uint8, device=dev)
    eng.set_linked_compress(True)
    eng.compress_batch_device(src, n, BL, slots, stride, flen, accel=accel)
    eng.compact_device(slots, stride, flen, n, dense, n * stride, doff)
    eng.synchronize()
    # my range's framed blocks: everything behind the look-back block (which only served as the dictionary)
    first = int(doff[lb].item())
    total = int(doff[-1].item())
    framed = dense[first:total]
    boff = (doff[lb:] - first).contiguous()
    mine = src[lb * BL:]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        if dist is None:
            ooff = torch.arange(NB + 1, dtype=torch.int64, device=dev) * BL
            out = torch.empty(NB * BL, dtype=torch.uint8, device=dev)
            res = torch.empty(NB, dtype=torch.int32, device=dev)
            eng.decompress_batch_device(framed, total - first, boff, NB, out, ooff, res, linked=True)
            eng.synchronize()
            return out, res
        return decode_linked_sharded(eng, framed, total - first, boff, [BL] * NB)

    out, res = step()
    if not (bool((res == BL).all().item()) and torch.equal(out, mine)):
        sys.exit("bench.py --one-stream: rank %d's range does not decode to the generator's blocks" % rank)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    cb = torch.tensor([float(total - first)], dtype=torch.float64, device=red_dev)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dist.all_reduce(cb, op=dist.ReduceOp.SUM)
        elapsed = float(tt.item())
    if rank == 0:
        U = NB * BL
        print(json.dumps({
            "metric": "GB/s uncompressed, ONE linked stream decoded by all ranks (not the headline metric)",
            "value": round(world * U / (elapsed / max(args.steps, 1)) / 1e9, 2), "unit": "GB/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / max(args.steps, 1), 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "one-stream: %s, %d KiB blocks, contiguous ranges of %d blocks per rank, ONE linked stream "
                                   "of %d blocks written by the engine's linked compression, seam block passed rank to rank"
                                   % (kind, BL >> 10, NB, NB * world),
                       "ratio": round(world * U / float(cb.item()), 4), "verified": True}}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def launch_ranks(n):
    """Run this script's command line under `python -m torch.distributed.run --nproc-per-node n` as a child process, relay
    rank 0's single JSON line (the ranks' other output goes to stderr) and return the child's exit code."""
    import socket
    with socket.socket() as sk:                     # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in child.stdout:
        if out.startswith("{") and line is None:
            line = out.rstrip("\n")
        else:
            sys.stderr.write(out)
    rc = child.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks printed no JSON line\n")
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="decompress", choices=sorted(WORKLOADS))
    ap.add_argument("--blocks", type=int, default=0, help="blocks per GPU (default: workload's)")
    ap.add_argument("--decoder", type=int, default=0, help="0 chosen per call, 1 sequence-at-a-time, 2 one wavefront per block, 4 one workgroup per block")
    ap.add_argument("--linked", action="store_true", help="decode with linked = 1 (reference stream semantics)")
    ap.add_argument("--linked-compress", action="store_true",
                    help="compress this rank's blocks as ONE linked stream (previous block = dictionary, like the "
                         "reference's compressor) and decode it with linked = 1")
    ap.add_argument("--one-stream", action="store_true",
                    help="the ranks' blocks are contiguous ranges of ONE linked stream (written by the engine's linked "
                         "compression), decoded range by range with the seam block passed from rank to rank "
                         "(streamly_lz4_amd/linked_shard.py); not the headline metric")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-all-cores", action="store_true", help="skip the multi-threaded best-case CPU figure")
    ap.add_argument("--cpu-sample-blocks", type=int, default=0)
    ap.add_argument("--no-gather", action="store_true", help="N>1: skip the RCCL ordered gather + its verification")
    ap.add_argument("--no-extra", action="store_true", help="skip the round-trip / compress figures measured after the timed steps")
    ap.add_argument("--no-host-api", action="store_true", help="skip the PCIe-inclusive host-buffer API figures (N=1)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` on its own: start the N ranks as a CHILD process (torch.distributed.run, one rank per
        # GPU) and hand back its exit code.  This process has not imported torch or touched the GPU, and it never
        # replaces itself: the ranks are children of the child.
        sys.exit(launch_ranks(args.gpus))

    import torch
    import streamly_lz4_amd as S

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        sys.exit("bench.py --gpus %d runs under WORLD_SIZE=%d: launch it with torch.distributed.run --nproc-per-node %d, "
                 "or without WORLD_SIZE set to let it start the ranks itself" % (args.gpus, world, args.gpus))
    dist = None
    # BENCH_FORCE_DEVICE / BENCH_DIST_BACKEND exist only to rehearse the N>1 code path on a 1-GPU box
    # (all ranks on one device, gloo rendezvous); the driver's launch uses one GPU per rank over RCCL.
    dev_index = int(os.environ.get("BENCH_FORCE_DEVICE", local_rank))
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    torch.cuda.set_device(dev_index)
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    dev = torch.device("cuda", dev_index)
    red_dev = dev if backend == "nccl" else torch.device("cpu")     # where the timing reduction lives

    kind, BL, NB, accel, phase = WORKLOADS[args.workload]
    if args.blocks:
        NB = args.blocks
    eng = S.Engine(dev_index)
    eng.set_decoder(args.decoder)
    if args.one_stream:
        one_stream_bench(args, torch, S, dist, eng, dev, rank, world, red_dev, kind, BL, NB, accel)
        return
    if args.linked_compress:
        eng.set_linked_compress(True)
        args.linked = True
        args.no_gather = True           # interleaving the ranks' streams would break the links
        args.no_host_api = True

    # ---- setup (untimed): this rank's blocks on the device, compressed and compacted ----
    U = NB * BL
    data_name = "synthetic"
    src = torch.empty(U, dtype=torch.uint8, device=dev)
    corpus_bytes = None
    if args.workload == "compress" and world == 1:
        corpus_bytes, cname = canterbury_large(U)
    if corpus_bytes is not None:
        import numpy as np
        src.copy_(torch.from_numpy(np.frombuffer(corpus_bytes, dtype=np.uint8).copy()))
        kind, data_name = "canterbury-large", cname
    else:
        eng.generate(kind, src, BL, NB, first_block=rank, block_step=world)      # block k -> rank k % N
    stride = S.slot_stride(BL, 8)
    slots = torch.empty(NB * stride, dtype=torch.uint8, device=dev)
    flen = torch.empty(NB, dtype=torch.int32, device=dev)
    doff = torch.empty(NB + 1, dtype=torch.int64, device=dev)
    dense = torch.empty(NB * stride, dtype=torch.uint8, device=dev)
    ooff = torch.arange(NB + 1, dtype=torch.int64, device=dev) * BL
    out = torch.empty(U, dtype=torch.uint8, device=dev)
    res = torch.empty(NB, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    def do_compress():
        eng.compress_batch_device(src, NB, BL, slots, stride, flen, accel=accel)
        eng.compact_device(slots, stride, flen, NB, dense, NB * stride, doff)

    def do_decompress():
        eng.decompress_batch_device(dense, Cbytes, doff, NB, out, ooff, res, linked=args.linked)

    Cbytes = NB * stride
    do_compress()                                                             # first touch
    eng.synchronize()
    Cbytes = int(doff[-1].item())                                             # compressed bytes incl. 8-byte headers
    do_decompress()
    eng.synchronize()
    if not (bool((res == BL).all().item()) and torch.equal(out, src)):
        sys.exit("bench.py: round trip mismatch during setup -- refusing to report a number")

    ev = [S.Event() for _ in range(4)]
    kern_ms = {"compress": [], "compact": [], "decompress": []}

    def step(ph):
        if ph in ("compress", "roundtrip"):
            eng.record(ev[0])
            eng.compress_batch_device(src, NB, BL, slots, stride, flen, accel=accel)
            eng.record(ev[1])
            eng.compact_device(slots, stride, flen, NB, dense, NB * stride, doff)
            eng.record(ev[2])
        if ph in ("decompress", "roundtrip"):
            if ph == "decompress":
                eng.record(ev[2])
            do_decompress()
            eng.record(ev[3])
        eng.synchronize()
        if ph in ("compress", "roundtrip"):
            kern_ms["compress"].append(eng.elapsed_ms(ev[0], ev[1]))
            kern_ms["compact"].append(eng.elapsed_ms(ev[1], ev[2]))
        if ph in ("decompress", "roundtrip"):
            kern_ms["decompress"].append(eng.elapsed_ms(ev[2], ev[3]))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step(phase)
    for k in kern_ms:
        kern_ms[k].clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(phase)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed * 1e3 / max(args.steps, 1)
    value = world * U / (elapsed / max(args.steps, 1)) / 1e9
    timed_ms = {k: (sum(v) / len(v) if v else None) for k, v in kern_ms.items()}

    # ---- after the timed steps: the other half of the metric's name (same data, same build) ----
    extra = None
    if not args.no_extra:
        for k in kern_ms:
            kern_ms[k].clear()
        barrier()
        r0 = time.perf_counter()
        n_rt = 3
        for _ in range(n_rt):
            step("roundtrip")
        barrier()
        rt_s = (time.perf_counter() - r0) / n_rt
        if dist is not None:
            tt = torch.tensor([rt_s], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            rt_s = float(tt.item())
        cm = sum(kern_ms["compress"]) / n_rt
        km = sum(kern_ms["compact"]) / n_rt
        dm = sum(kern_ms["decompress"]) / n_rt
        extra = {"roundtrip_GBps": round(world * U / rt_s / 1e9, 2), "roundtrip_ms_per_step": round(rt_s * 1e3, 4),
                 "kernels_ms": {"compress": round(cm, 4), "compact": round(km, 4), "decompress": round(dm, 4)},
                 "compress": {"GBps": round(U / cm / 1e6, 2), "achieved": round((U + Cbytes) / cm / 1e6, 2),
                              "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                              "frac": round((U + Cbytes) / cm / 1e6 / HBM_PEAK_GBPS, 5), "avg_launch_ms": round(cm, 4)},
                 "decompress": {"GBps": round(U / dm / 1e6, 2), "frac": round((U + Cbytes) / dm / 1e6 / HBM_PEAK_GBPS, 5)}}

    # ---- N>1: ordered RCCL gather of the framed output to rank 0, verified, reported separately ----
    gather = None
    if dist is not None and not args.no_gather:
        from streamly_lz4_amd.gather import gather_ordered
        do_compress()
        eng.synchronize()
        fl = flen.clone()
        # `value` does not depend on the gather: a failure in it (it is the only place where ranks exchange data) is
        # reported in the line instead of taking the measurement down with it
        try:
            gathered, goff = gather_ordered(dense[:Cbytes], fl, root=0, engine=eng)     # untimed: sets up the peer connections
            del gathered
            barrier()
            g0 = time.perf_counter()
            gathered, goff = gather_ordered(dense[:Cbytes], fl, root=0, engine=eng)
            barrier()
            gather_s = time.perf_counter() - g0
            gather = {"ms": round(gather_s * 1e3, 3)}
            if rank == 0:
                # the gathered stream must decode to the generator's GLOBAL stream (block k = generator block k)
                nG = NB * world
                gsrc = torch.empty(nG * BL, dtype=torch.uint8, device=dev)
                eng.generate(kind, gsrc, BL, nG, first_block=0, block_step=1)
                gout = torch.empty(nG * BL, dtype=torch.uint8, device=dev)
                gres = torch.empty(nG, dtype=torch.int32, device=dev)
                gooff = torch.arange(nG + 1, dtype=torch.int64, device=dev) * BL
                eng.decompress_batch_device(gathered, int(goff[-1].item()), goff, nG, gout, gooff, gres)
                eng.synchronize()
                ok = bool((gres == BL).all().item()) and torch.equal(gout, gsrc)
                gather["verified"] = ok
                gather["bytes"] = int(goff[-1].item())
                if not ok:
                    # `value` does not depend on the gather: the line still goes out, and says that the gather is wrong
                    gather["error"] = "the gathered stream does not decode to the generator's global stream"
                    sys.stderr.write("bench.py: " + gather["error"] + "\n")
                del gsrc, gout, gres, gooff
            del gathered
        except SystemExit:
            raise
        except Exception as e:                                            # noqa: BLE001
            gather = {"error": ("%s: %s" % (type(e).__name__, e))[:300]}

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel of the timed phase ----
    dom = "decompress" if phase in ("decompress", "roundtrip") else "compress"
    avg_ms = timed_ms[dom] or 0.0
    achieved = (U + Cbytes) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic, traffic_src = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            # the PMC figure is per launch of the workload's full-size batch
            traffic = tj.get("%s:%s" % (args.workload, dom)) if NB == WORKLOADS[args.workload][2] and corpus_bytes is None else None
            tcommit = (tj.get("_commits") or {}).get("%s:%s" % (args.workload, dom)) or tj.get("_commit", "an earlier commit")
            traffic_src = ("profiles/traffic.json: rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, one pass each) of this command "
                           "at commit %s -- a constant read back here, not measured by this run" % tcommit)
        except Exception:
            traffic = None
    # what a plain device-to-device copy of the same U bytes reaches on this box (read + write)
    copy_GBps = None
    try:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        out.copy_(src); torch.cuda.synchronize()
        ev0.record(); out.copy_(src); ev1.record(); torch.cuda.synchronize()
        copy_GBps = round(2 * U / (ev0.elapsed_time(ev1) * 1e-3) / 1e9, 1)
    except Exception:
        copy_GBps = None
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": U + Cbytes, "avg_launch_ms": round(avg_ms, 4),
                "device_copy_GBps": copy_GBps}

    # ---- CPU baseline on a bounded sample of the same input (rank 0, N=1 only) ----
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        ns = args.cpu_sample_blocks or min(NB, (1 << 30) // BL)                # <= 1 GiB of the same stream
        host = src[: ns * BL].cpu().numpy()
        blocks = [host[i * BL:(i + 1) * BL].tobytes() for i in range(ns)]
        r = orc.cpu_baseline(blocks, accel=accel, keep_stream=True)
        cpu_dec = r["raw_bytes"] / r["decomp_s"] / 1e9
        cpu_cmp = r["raw_bytes"] / r["comp_s"] / 1e9
        gpu_sample_bytes = int(doff[ns].item())
        cpu = {"value": round(cpu_dec if dom == "decompress" else cpu_cmp, 3), "unit": "GB/s", "cores": 1,
               "kind": r["kind"],
               "sample": "first %d blocks (%d MiB) of the same %s stream, reference call sequence (one linked context), best of 3"
                         % (ns, ns * BL >> 20, kind),
               "decompress_GBps": round(cpu_dec, 3), "compress_GBps": round(cpu_cmp, 3),
               "ratio": round(r["raw_bytes"] / (r["comp_bytes"] + 8 * ns), 4),
               "gpu_ratio_same_sample": round(ns * BL / gpu_sample_bytes, 4),
               "gpu_size_vs_reference": round(gpu_sample_bytes / (r["comp_bytes"] + 8 * ns), 4)}
        # The stream the reference wrote for this sample (one linked context: nearly every block needs the output of
        # the block before it) decoded by the GPU in one call, linked = 1, and compared with the input.
        cpu["reference_stream_gpu_decode"] = reference_stream_decode(S, eng, torch, dev, r["stream"], src, ns, BL)
        if ns * BL <= (1 << 30) and torch.cuda.mem_get_info(dev)[0] > 12 * ns * BL:
            cpu["reference_stream_gpu_decode_x4"] = reference_stream_decode(S, eng, torch, dev, r["stream"], src, ns, BL, copies=4)
        del r["stream"]
        # ... and the same sample compressed block by block with a fresh reference context (independent blocks): what the
        # timed kernel does on a stream it did not write itself
        codec = orc.Reference() if r["kind"] == "reference" else orc.Oracle()
        indep = [codec.compress_block(b, accel) for b in blocks]
        cpu["decode_reference_written_independent"] = reference_stream_decode(S, eng, torch, dev, indep, src, ns, BL, linked=False)
        del indep
        # ... and, for a like-for-like reading of that figure, the ENGINE's own stream of the same sample in a call of the same
        # size: a call of 16 384 blocks decodes 10 % slower than `value`'s 65 536 whoever wrote it (the last waves of a launch
        # run on a GPU that is emptying), the writer is worth about 1 % (scripts/par_stats_ref.py)
        if dom == "decompress" and ns < NB:
            e0, e1 = S.Event(), S.Event()
            best = 1e9
            for _ in range(3):
                eng.record(e0)
                eng.decompress_batch_device(dense, gpu_sample_bytes, doff, ns, out, ooff, res)
                eng.record(e1)
                eng.synchronize()
                best = min(best, eng.elapsed_ms(e0, e1))
            cpu["decode_reference_written_independent"]["engine_written_same_sample_GBps"] = round(ns * BL / best / 1e6, 2)
        if not args.no_cpu_all_cores:
            # best-case CPU, NOT reference behaviour (its API is one serial stream): one independent
            # linked context per host thread over contiguous block ranges of the same sample
            ra = orc.cpu_baseline_all_cores(blocks, accel=accel)
            cpu["all_cores"] = {"cores": ra["threads"], "decompress_GBps": round(ra["raw_bytes"] / ra["decomp_s"] / 1e9, 3),
                                "compress_GBps": round(ra["raw_bytes"] / ra["comp_s"] / 1e9, 3),
                                "note": "best-case CPU, not reference behaviour: one stream per thread"}

    line = {
        "metric": "GB/s uncompressed (compress+decompress), 64 KiB blocks, 1/2/4/8 GPU",
        "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u8", "data": data_name,
        "config": {"workload": "%s: %s, %d KiB blocks, %d blocks (%.2f GiB) per GPU, %s, accel %d, %s, "
                               "round-robin block->GPU%s" % (args.workload, phase, BL >> 10, NB, U / 2 ** 30, kind, accel,
                                                             "ONE LINKED stream per GPU (previous block = dictionary), written by this engine"
                                                             if args.linked_compress else
                                                             "independent blocks WRITTEN BY THIS ENGINE's compressor during setup",
                                                             ", linked=1" if args.linked else ""),
                   "block_len": BL, "blocks_per_gpu": NB, "ratio": round(U / Cbytes, 4), "decoder": args.decoder},
        "roofline": roofline,
        "cpu_baseline": cpu,
        "kernels_ms": {k: round(v, 4) for k, v in timed_ms.items() if v},
    }
    if cpu is not None and "decode_reference_written_independent" in cpu:
        line["decode_reference_written_independent_GBps"] = cpu["decode_reference_written_independent"]["GBps"]
    if extra is not None:
        line["roundtrip"] = extra
    if world == 1 and not args.no_extra and args.workload != "random256k" and corpus_bytes is None:
        line["incompressible"] = incompressible_rates(S, eng, torch, dev)
    if world == 1 and not args.no_host_api and kind != "canterbury-large":
        line["host_api_pcie_inclusive"] = host_api_rates(S, eng, src, BL, kind)
    if world == 1 and not args.no_extra and not args.linked_compress and kind != "canterbury-large":
        line["small_call"] = small_call_rates(S, eng, torch, dev, src, BL, kind, accel, not args.no_cpu_baseline)
    gather_failed = False
    if gather is not None:
        # compute-only is `value` (N x one GPU by construction: no collective in the timed region).  The numbers that
        # test the multi-GPU design are the gather's own: what the root RECEIVES (the peers' framed bytes: (N - 1) / N of
        # the gathered stream) over its time, against the root's seven xGMI links (7 x 153 GB/s, MI355X_MICROARCH.md), and the
        # round trip with one ordered gather per pass added to it.
        cm = (extra or {}).get("kernels_ms", {}).get("compress")
        if "ms" in gather and gather.get("bytes"):
            recv = gather["bytes"] * (world - 1) / world
            gather["received_bytes"] = int(recv)
            gather["GBps"] = round(recv / gather["ms"] / 1e6, 2)
            gather["xgmi_peak_GBps"] = 7 * 153.0
            gather["frac_of_xgmi_peak"] = round(recv / gather["ms"] / 1e6 / (7 * 153.0), 4)
        if cm and "ms" in gather:
            gather["compress_only_GBps"] = round(world * U / cm / 1e6, 2)
            gather["compress_plus_gather_GBps"] = round(world * U / (cm + gather["ms"]) / 1e6, 2)
        if extra is not None and "ms" in gather:
            line["roundtrip_plus_gather_GBps"] = round(world * U / (extra["roundtrip_ms_per_step"] + gather["ms"]) / 1e6, 2)
        try:
            gather["backend"] = dist.get_backend()              # "nccl" = RCCL on ROCm; "gloo" = a rehearsal without xGMI
        except Exception:
            pass
        line["gather"] = gather
        if gather.get("error") or gather.get("verified") is False:
            line["gather_verified"] = False     # (`value` is verified on its own: the gather is not in the timed region)
            # A gather that COMPLETED and delivered a stream that does not decode to the global input fails the run (exit
            # code 1 below): the ordered gather is then wrong.  A gather that could not run at all (an exception out of the
            # communication layer) is reported in the line and leaves the exit code alone: `value` does not depend on it.
            gather_failed = gather.get("verified") is False
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if gather_failed:
        sys.exit(1)


if __name__ == "__main__":
    main()
