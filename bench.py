#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X LZ4 block engine.

    python bench.py --gpus N --steps K --warmup W [--workload decompress|compress|roundtrip|random256k]

One "step" = one pass of the hot path over one batch of synthetic input resident in HBM.
Default workload (N=1) is BASELINE.json configs[1]: decompress-only, 64 KiB blocks, 4 GiB
lzsynth(16, 2048) stream of independent blocks (compressed by this engine during setup).  For N>1
(one process per GPU under torch.distributed.run, RCCL) block k of the global stream lives on rank
k % N (per-block round-robin); every rank processes 4 GiB (weak scaling) with no data-path
collective.  Rank 0 prints ONE JSON line.

`value` is whole-job GB/s of UNCOMPRESSED bytes over the wall-clock of the K timed steps (max over
ranks).  `roofline` prices the dominant kernel: algorithmic bytes (U + C) per launch /
its average launch duration, measured live with HIP events on the engine's own stream, against the
8 TB/s HBM peak.  `cpu_baseline` times the reference codec (oracle/_ref, kind "reference"; or the
oracle port) on ONE host core on a bounded sample of the same input -- context, not the target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    # name: (kind, block_len, n_blocks_per_gpu, accel, timed phase)
    "decompress": ("lzsynth", 65536, 65536, 1, "decompress"),   # BASELINE configs[1]
    "compress": ("lzsynth", 65536, 65536, 1, "compress"),       # configs[2] shape on lzsynth (Canterbury absent offline)
    "roundtrip": ("lzsynth", 65536, 65536, 1, "roundtrip"),     # configs[3] per-GPU share
    "random256k": ("random", 262144, 16384, 400, "roundtrip"),  # configs[4] per-GPU share
    "text": ("text", 65536, 65536, 1, "decompress"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="decompress", choices=sorted(WORKLOADS))
    ap.add_argument("--blocks", type=int, default=0, help="blocks per GPU (default: workload's)")
    ap.add_argument("--decoder", type=int, default=0, help="0 auto, 1 sequence-at-a-time, 2 lane-parallel")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-all-cores", action="store_true", help="skip the multi-threaded best-case CPU figure")
    ap.add_argument("--cpu-sample-blocks", type=int, default=0)
    ap.add_argument("--gather", action="store_true", help="also time the RCCL ordered gather of the framed output (N>1)")
    args = ap.parse_args()

    import torch
    import streamly_lz4_amd as S

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1):
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
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

    # ---- setup (untimed): generate this rank's blocks on the device, compress, compact ----
    U = NB * BL
    src = torch.empty(U, dtype=torch.uint8, device=dev)
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
        eng.decompress_batch_device(dense, Cbytes, doff, NB, out, ooff, res)

    Cbytes = NB * stride
    do_compress()                                                             # first touch
    eng.synchronize()
    se = [S.Event() for _ in range(3)]
    eng.record(se[0])
    eng.compress_batch_device(src, NB, BL, slots, stride, flen, accel=accel)
    eng.record(se[1])
    eng.compact_device(slots, stride, flen, NB, dense, NB * stride, doff)
    eng.record(se[2])
    eng.synchronize()
    setup_compress_ms = eng.elapsed_ms(se[0], se[1])                          # reported as context (untimed phase)
    setup_compact_ms = eng.elapsed_ms(se[1], se[2])
    Cbytes = int(doff[-1].item())                                             # compressed bytes incl. 8-byte headers
    do_decompress()
    eng.synchronize()
    if not (bool((res == BL).all().item()) and torch.equal(out, src)):
        sys.exit("bench.py: round trip mismatch during setup -- refusing to report a number")

    ev = [S.Event() for _ in range(4)]
    kern_ms = {"compress": [], "compact": [], "decompress": []}

    def step():
        if phase in ("compress", "roundtrip"):
            eng.record(ev[0])
            eng.compress_batch_device(src, NB, BL, slots, stride, flen, accel=accel)
            eng.record(ev[1])
            eng.compact_device(slots, stride, flen, NB, dense, NB * stride, doff)
            eng.record(ev[2])
        if phase in ("decompress", "roundtrip"):
            if phase == "decompress":
                eng.record(ev[2])
            do_decompress()
            eng.record(ev[3])
        eng.synchronize()
        if phase in ("compress", "roundtrip"):
            kern_ms["compress"].append(eng.elapsed_ms(ev[0], ev[1]))
            kern_ms["compact"].append(eng.elapsed_ms(ev[1], ev[2]))
        if phase in ("decompress", "roundtrip"):
            kern_ms["decompress"].append(eng.elapsed_ms(ev[2], ev[3]))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    for k in kern_ms:
        kern_ms[k].clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed * 1e3 / max(args.steps, 1)
    value = world * U / (elapsed / max(args.steps, 1)) / 1e9

    # ---- optional: ordered RCCL gather of the framed output to rank 0 (reported separately) ----
    gather_ms = None
    if args.gather and dist is not None:
        from streamly_lz4_amd.gather import gather_ordered
        fl = flen.clone()
        barrier()
        g0 = time.perf_counter()
        gathered, goff = gather_ordered(dense[:Cbytes], fl, root=0, engine=eng)
        barrier()
        gather_ms = (time.perf_counter() - g0) * 1e3
        del gathered

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel ----
    dom = "decompress" if phase in ("decompress", "roundtrip") else "compress"
    avg_ms = sum(kern_ms[dom]) / max(len(kern_ms[dom]), 1)
    achieved = (U + Cbytes) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get("%s:%s" % (args.workload, dom))
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
                "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                "algorithmic_bytes_per_launch": U + Cbytes, "avg_launch_ms": round(avg_ms, 4),
                "device_copy_GBps": copy_GBps}

    # ---- CPU baseline on a bounded sample of the same input (rank 0, N=1 only) ----
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc
        ns = args.cpu_sample_blocks or min(NB, (1 << 30) // BL)                # <= 1 GiB of the same stream
        host = src[: ns * BL].cpu().numpy()
        blocks = [host[i * BL:(i + 1) * BL].tobytes() for i in range(ns)]
        r = orc.cpu_baseline(blocks, accel=accel)
        cpu_dec = r["raw_bytes"] / r["decomp_s"] / 1e9
        cpu_cmp = r["raw_bytes"] / r["comp_s"] / 1e9
        cpu = {"value": round(cpu_dec if dom == "decompress" else cpu_cmp, 3), "unit": "GB/s", "cores": 1,
               "kind": r["kind"],
               "sample": "first %d blocks (%d MiB) of the same %s stream, reference call sequence (one linked context), best of 3"
                         % (ns, ns * BL >> 20, kind),
               "decompress_GBps": round(cpu_dec, 3), "compress_GBps": round(cpu_cmp, 3),
               "ratio": round(r["raw_bytes"] / (r["comp_bytes"] + 8 * ns), 4)}
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
        "dtype": "u8", "data": "synthetic",
        "config": {"workload": "%s: %s, %d KiB blocks, %d blocks (%.2f GiB) per GPU, %s, accel %d, independent blocks, "
                               "round-robin block->GPU" % (args.workload, phase, BL >> 10, NB, U / 2 ** 30, kind, accel),
                   "block_len": BL, "blocks_per_gpu": NB, "ratio": round(U / Cbytes, 4), "decoder": args.decoder},
        "roofline": roofline,
        "cpu_baseline": cpu,
        "kernels_ms": {k: round(sum(v) / len(v), 4) for k, v in kern_ms.items() if v},
        # context from the untimed setup pass over the same data (per GPU): the other half of the metric's name
        "setup": {"compress_GBps": round(U / setup_compress_ms / 1e6, 2), "compact_ms": round(setup_compact_ms, 4),
                  "compress_roofline_frac": round((U + Cbytes) / setup_compress_ms / 1e6 / HBM_PEAK_GBPS, 5)},
    }
    if gather_ms is not None:
        line["gather_ms"] = round(gather_ms, 3)
    print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
