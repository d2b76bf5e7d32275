"""Development aid: phase breakdown of the encoder's dense-window path (ENC_STATS build via MI355LZ4_LIB)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "streamly-lz4_amd"))
import torch, streamly_lz4_amd as S
kind = sys.argv[1] if len(sys.argv) > 1 else "lzsynth"
linked = len(sys.argv) > 2 and sys.argv[2] == "linked"
NB = 8192; BL = 65536
dev = torch.device("cuda:0"); eng = S.Engine(0); eng.set_linked_compress(linked)
S.lib.mi355lz4_debug_stats.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]
src = torch.empty(NB * BL, dtype=torch.uint8, device=dev); eng.generate(kind, src, BL, NB)
stride = S.slot_stride(BL, 8)
slots = torch.empty(NB * stride, dtype=torch.uint8, device=dev); flen = torch.empty(NB, dtype=torch.int32, device=dev)
buf = (C.c_uint64 * 32)()
S.lib.mi355lz4_debug_stats(eng.ctx, 1, buf)
eng.compress_batch_device(src, NB, BL, slots, stride, flen); eng.synchronize()
S.lib.mi355lz4_debug_stats(eng.ctx, 0, buf)
v = list(buf)[:8]; w = max(v[6], 1)
names = ["0:probe+heads+requests issued", "1:wait+lengths+select", "2:next bytes requested", "3:finish+queue moves+table", "4:wait for the windows bytes", "5:hash + table round trip", "windows", "heads"]
print(kind, "linked" if linked else "independent", "windows/block %.0f" % (w / NB), " cycles/window:", {n: round(x / w) for n, x in zip(names, v) if n != "windows"})
