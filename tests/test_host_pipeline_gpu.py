"""The pipelined host-buffer calls with MANY groups per call: MI355LZ4_GROUP_MB=1 makes a
24 MiB call flow through ~24 groups on the three streams.  Run in a child process because the group size is
read once per process.  Checked against the oracle: round trip, the reference's linked stream (every block
reaching into its predecessor, across group boundaries), blocks that decode short of their capacity, a
corrupted block, ragged blocks, page-locked caller buffers, and linked compression across group boundaries."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
ROOT = %r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import ctypes as C
import numpy as np, torch
import streamly_lz4_amd as S
from oracle.oracle import Oracle
O = Oracle(); eng = S.Engine(0)
bl, n = 65536, 384
raw = O.gen("text", n, bl).tobytes()
blocks = [raw[i * bl:(i + 1) * bl] for i in range(n)]
# 1. compress (24 groups) -> oracle decodes; decompress (24 groups) -> identity
fr, flen = eng.compress_batch(blocks, accel=1)
assert len(fr) == sum(flen) and O.frame_decompress(fr, n * bl, 8, 0, True) == raw
out, blen = eng.decompress_batch(fr)
assert blen == [bl] * n and out == raw
# 2. the reference's LINKED stream: dependencies cross every group boundary
ref = O.frame_compress(raw, bl, 1, 8, True)
out, blen = eng.decompress_batch(ref, linked=True)
assert blen == [bl] * n and out == raw
out, blen = eng.decompress_batch(ref, linked=False, raise_on_block_error=False)
assert sum(1 for b in blen if b < 0) > n // 2                      # standalone: most blocks fail, with codes
# 3. linked with a dictionary in force before block 0
half = n // 2
tail = b"".join(blocks[half:])
pos = 0
for _ in range(half):
    pos += 8 + int.from_bytes(ref[pos:pos + 4], "little")
out, blen = eng.decompress_batch(ref[pos:], linked=True, dict_bytes=blocks[half - 1])
assert blen == [bl] * (n - half) and out == tail
# 4. capacity larger than the decoded size (BlockMax256KB-style fixed capacity): blocks pack back to back
small = [raw[i * 5000:(i + 1) * 5000] for i in range(300)]
fr4 = b"".join(len(c).to_bytes(4, "little") + c for c in (O.compress_block(b, 1) for b in small))
out, blen = eng.decompress_batch(fr4, header_kind=4, fixed_uncomp=262144)
assert blen == [5000] * 300 and out == b"".join(small)
# 5. one corrupted block: its code, the others untouched
bad = bytearray(fr)
off = sum(flen[:100])
bad[off + 8 + 20:off + 8 + 40] = bytes(20)
want = O.decompress_block(bytes(bad[off + 8:off + flen[100]]), bl)[0]
out, blen = eng.decompress_batch(bytes(bad), raise_on_block_error=False)
assert blen[100] == want and all(b == bl for i, b in enumerate(blen) if i != 100)
# 6. ragged blocks incl. empty ones
rag = [raw[:0], raw[:1], raw[:13], raw[100:70000], raw[:0], raw[5:200000]] * 20
fr6, fl6 = eng.compress_batch(rag, accel=3)
out, blen = eng.decompress_batch(fr6)
assert blen == [len(b) for b in rag] and out == b"".join(rag)
# 7. page-locked caller buffers go to the DMA engines directly
u8p, i32p = C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
host = torch.from_numpy(np.frombuffer(raw, dtype=np.uint8).copy()).pin_memory()
cap = n * (S.compress_bound(bl) + 8)
framed = torch.empty(cap, dtype=torch.uint8).pin_memory(); back = torch.empty(n * bl, dtype=torch.uint8).pin_memory()
ptrs = (u8p * n)(*[C.cast(host.data_ptr() + i * bl, u8p) for i in range(n)])
lens = np.full(n, bl, dtype=np.int32); fl = np.zeros(n, dtype=np.int32); st = np.zeros(n, dtype=np.int32)
olen = C.c_size_t(); dlen = C.c_size_t(); got = C.c_int()
assert S.lib.mi355lz4_compress_batch(eng.ctx, ptrs, lens.ctypes.data_as(i32p), n, 1, 8, C.cast(framed.data_ptr(), u8p), cap,
                                     C.byref(olen), fl.ctypes.data_as(i32p), st.ctypes.data_as(i32p)) == 0
assert bytes(framed.numpy()[:olen.value]) == fr                      # same bytes as through the staged path
assert S.lib.mi355lz4_decompress_batch(eng.ctx, C.cast(framed.data_ptr(), u8p), olen.value, 8, 0, 1, None, 0,
                                       C.cast(back.data_ptr(), u8p), n * bl, C.byref(dlen), fl.ctypes.data_as(i32p), n, C.byref(got)) == 0
assert dlen.value == n * bl and bytes(back.numpy()) == raw
# 8. linked compression through many groups: a group's first block has the last block of the group before as its
#    dictionary (every block but the first reaches back), both linked decoders return the input
eng.set_linked_compress(True)
frl, fll = eng.compress_batch(blocks, accel=1)
eng.set_linked_compress(False)
assert len(frl) < 0.97 * len(fr) and O.frame_decompress(frl, n * bl, 8, 0, True) == raw
out, blen = eng.decompress_batch(frl, linked=True)
assert blen == [bl] * n and out == raw
out, blen = eng.decompress_batch(frl, linked=False, raise_on_block_error=False)
assert blen[0] == bl and sum(1 for b in blen[1:] if b < 0) >= n - 8
print("pipeline ok")
'''


@pytest.mark.parametrize("group_mb", ["1", "64"])
def test_host_api_pipelined_groups(group_mb):
    env = dict(os.environ, MI355LZ4_GROUP_MB=group_mb)
    r = subprocess.run([sys.executable, "-c", CHILD % ROOT], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "pipeline ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
