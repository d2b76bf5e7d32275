"""Development aid (GPU box): a linked stream of 1 MiB blocks through the host-buffer call (groups of 64 MiB: the groups behind the
first have blocks in front of them), with and without the big-block path.   python scripts/big_linked_host.py [blocks]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import numpy as np, streamly_lz4_amd as S
from oracle.oracle import Oracle
O = Oracle(); eng = S.Engine(0)
bl = 1 << 20; nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 200
raw = O.gen("text", nblk * bl // 65536, 65536, first_block=77).tobytes()[: nblk * bl - 12345]
fr = O.frame_compress(raw, bl, 1, 8, True)
for env in (None, "0"):
    if env is None: os.environ.pop("MI355LZ4_LINKED_BIG", None)
    else: os.environ["MI355LZ4_LINKED_BIG"] = env
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter(); out, blen = eng.decompress_batch(fr, linked=True); best = min(best, time.perf_counter() - t0)
    print("host call, %d linked blocks of 1 MiB: LINKED_BIG=%s  %.2f ms  %.1f GB/s  ok %s" % (nblk, env, best * 1e3, len(raw) / best / 1e9, out == raw and blen == [bl] * (nblk - 1) + [bl - 12345]), flush=True)
