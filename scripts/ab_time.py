"""Development aid: time decode (and optionally encode) of every library variant in one process tree.
    python scripts/ab_time.py [kinds] [nblocks]   (runs each variant in a subprocess via MI355LZ4_LIB)"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
kinds = sys.argv[1] if len(sys.argv) > 1 else "lzsynth,text"
nb = sys.argv[2] if len(sys.argv) > 2 else "32768"
child = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "streamly-lz4_amd"))
import torch, streamly_lz4_amd as S
dev = torch.device("cuda:0"); eng = S.Engine(0); BL = 65536; NB = int(sys.argv[2])
out_line = []
for kind in sys.argv[1].split(","):
    src = torch.empty(NB * BL, dtype=torch.uint8, device=dev); eng.generate(kind, src, BL, NB)
    stride = S.slot_stride(BL, 8)
    slots = torch.empty(NB * stride, dtype=torch.uint8, device=dev); flen = torch.empty(NB, dtype=torch.int32, device=dev)
    dense = torch.empty(NB * stride, dtype=torch.uint8, device=dev); doff = torch.empty(NB + 1, dtype=torch.int64, device=dev)
    ooff = torch.arange(NB + 1, dtype=torch.int64, device=dev) * BL
    out = torch.empty(NB * BL, dtype=torch.uint8, device=dev); res = torch.empty(NB, dtype=torch.int32, device=dev)
    e = [S.Event() for _ in range(3)]
    tc = td = 1e9
    for it in range(4):
        eng.record(e[0]); eng.compress_batch_device(src, NB, BL, slots, stride, flen)
        eng.record(e[1]); eng.compact_device(slots, stride, flen, NB, dense, NB * stride, doff); eng.synchronize()
        tc = min(tc, eng.elapsed_ms(e[0], e[1]))
    tds = {}
    ok = True
    for dv in (2,):
        eng.set_decoder(dv); td = 1e9
        out.zero_()
        for it in range(6):
            eng.record(e[1]); eng.decompress_batch_device(dense, NB * stride, doff, NB, out, ooff, res); eng.record(e[2]); eng.synchronize()
            td = min(td, eng.elapsed_ms(e[1], e[2]))
        tds[dv] = td
        ok = ok and bool((res == BL).all().item()) and torch.equal(out, src)
    C = int(doff[-1].item()); U = NB * BL
    out_line.append("%%s: dec %%.0f GB/s (U+C %%.0f) enc %%.0f GB/s ratio %%.3f ok=%%s" %% (kind, U / tds[2] / 1e6, (U + C) / tds[2] / 1e6, U / tc / 1e6, U / C, ok))
print(" | ".join(out_line))
''' % (ROOT, ROOT)
main = os.path.join(ROOT, "streamly-lz4_amd", "lib", "libmi355lz4.so")
# the working tree's build runs first AND last: the first process of a session on a fresh box measures 1-2 % low
libs = [main] + sorted(glob.glob(os.path.join(ROOT, "streamly-lz4_amd", "lib", "variants", "*.so"))) + [main]
for lib in libs:
    env = dict(os.environ, MI355LZ4_LIB=lib)
    r = subprocess.run([sys.executable, "-c", child, kinds, nb], env=env, capture_output=True, text=True)
    print("%-28s %s" % (os.path.basename(lib), (r.stdout.strip().splitlines() or [r.stderr.strip()[-300:]])[-1]), flush=True)
