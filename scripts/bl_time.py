"""Development aid: encode + decode rate of blocks of a given size on every library variant (subprocess per variant via MI355LZ4_LIB).
    python scripts/bl_time.py [block_len=262144] [blocks=8192] [kinds=lzsynth,text]"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bl = sys.argv[1] if len(sys.argv) > 1 else "262144"; nb = sys.argv[2] if len(sys.argv) > 2 else "8192"
kinds = sys.argv[3] if len(sys.argv) > 3 else "lzsynth,text"
child = r'''
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, "streamly-lz4_amd"))
import torch, streamly_lz4_amd as S
dev = torch.device("cuda:0"); eng = S.Engine(0); BL = int(sys.argv[2]); NB = int(sys.argv[3]); out_line = []
for kind in sys.argv[1].split(","):
    src = torch.empty(NB * BL, dtype=torch.uint8, device=dev); eng.generate(kind, src, 65536, NB * BL // 65536)
    stride = S.slot_stride(BL, 8)
    slots = torch.empty(NB * stride, dtype=torch.uint8, device=dev); flen = torch.empty(NB, dtype=torch.int32, device=dev)
    e = [S.Event() for _ in range(2)]; tc = 1e9
    for it in range(4):
        eng.record(e[0]); eng.compress_batch_device(src, NB, BL, slots, stride, flen); eng.record(e[1]); eng.synchronize()
        tc = min(tc, eng.elapsed_ms(e[0], e[1]))
    out_line.append("%%s: enc %%.1f GB/s ratio %%.3f" %% (kind, NB * BL / tc / 1e6, NB * BL / float(flen.sum().item())))
print(" | ".join(out_line))
''' % (ROOT, ROOT)
main = os.path.join(ROOT, "streamly-lz4_amd", "lib", "libmi355lz4.so")
for lib in [main] + sorted(glob.glob(os.path.join(ROOT, "streamly-lz4_amd", "lib", "variants", "*.so"))) + [main]:
    r = subprocess.run([sys.executable, "-c", child, kinds, bl, nb], env=dict(os.environ, MI355LZ4_LIB=lib), capture_output=True, text=True)
    print("%-28s %s" % (os.path.basename(lib), (r.stdout.strip().splitlines() or [r.stderr.strip()[-300:]])[-1]), flush=True)
