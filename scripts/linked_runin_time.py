"""Development aid: ONE reference-written linked stream (oracle compressor on the host), decoded by the GPU with the run-in
decode off / on and different run-in and piece lengths (RUNIN_CFGS="blocks of run-in:blocks per piece,...", 0 = default);
every output is compared with the input.
usage: linked_runin_time.py [kind=text] [blocks=16384] [block_len=65536]"""
import os, sys, struct, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import numpy as np, torch, streamly_lz4_amd as S
from oracle.oracle import Oracle
kind = sys.argv[1] if len(sys.argv) > 1 else "text"
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
bl = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
O = Oracle(); eng = S.Engine(0); dev = torch.device("cuda:0")
t0 = time.time()
# a few hundred distinct blocks cycled with a stride keep the host compression short; the stream is still one linked stream
base_n = nb if os.environ.get("RUNIN_DISTINCT") else min(nb, 2048)
data = O.gen(kind, base_n, bl, first_block=7).tobytes()
data = (data * ((nb + base_n - 1) // base_n))[: nb * bl]
fr = O.frame_compress(data, bl, 1, 8, True)
offs = np.zeros(nb + 1, dtype=np.int64); pos = 0
for i in range(nb):
    offs[i] = pos; pos += 8 + int.from_bytes(fr[pos:pos + 4], "little")
offs[nb] = pos
print("%s: %d blocks of %d, ratio %.3f, host side %.1f s" % (kind, nb, bl, nb * bl / len(fr), time.time() - t0), flush=True)
buf = torch.from_numpy(np.frombuffer(fr, dtype=np.uint8).copy()).to(dev)
off = torch.from_numpy(offs).to(dev)
ooff = torch.arange(nb + 1, dtype=torch.int64, device=dev) * bl
src = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).to(dev)
out = torch.zeros(nb * bl, dtype=torch.uint8, device=dev); res = torch.zeros(nb, dtype=torch.int32, device=dev)
e0, e1 = S.Event(), S.Event()
runins = [x.split(":") for x in os.environ.get("RUNIN_CFGS", "").split(",") if x]      # "blocks of run-in:blocks per piece" (0 = default)
configs = [("pointer pass", {"MI355LZ4_LINKED_RUNIN": "0"})]
for w, b in runins:
    env = {"MI355LZ4_LINKED_RUNIN": "1"}
    if int(w): env["MI355LZ4_LINKED_RUNIN_BLOCKS"] = w
    if int(b): env["MI355LZ4_LINKED_RUNIN_PIECE"] = b
    configs.append(("run-in %s, pieces of %s" % (w, b), env))
configs.append(("default", {}))
for label, env in configs:
    if os.environ.get("RUNIN_ONLY") and os.environ["RUNIN_ONLY"] not in label:
        continue
    for k in ("MI355LZ4_LINKED_RUNIN", "MI355LZ4_LINKED_RUNIN_BLOCKS", "MI355LZ4_LINKED_RUNIN_PIECE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    best = 1e9
    for _ in range(3):
        out.zero_(); res.zero_()
        eng.record(e0); eng.decompress_batch_device(buf, len(fr), off, nb, out, ooff, res, linked=True); eng.record(e1); eng.synchronize()
        best = min(best, eng.elapsed_ms(e0, e1))
    ok = bool((res == bl).all().item()) and torch.equal(out, src)
    print("  %-26s %8.3f ms  %7.1f GB/s  %s" % (label, best, nb * bl / best / 1e6, "ok" if ok else "MISMATCH"), flush=True)
