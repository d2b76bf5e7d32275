#!/usr/bin/env python3
"""Development aid: static instruction counts per source line of one kernel.
usage: isa_lines.py kernels.s <mangled-kernel-name> [file-substring]
(kernels.s from: hipcc -O3 -std=c++17 --offload-arch=gfx950 -gline-tables-only -S --cuda-device-only)"""
import re, sys, collections
path, fn = sys.argv[1], sys.argv[2]
want = sys.argv[3] if len(sys.argv) > 3 else None
files = {}
cnt = collections.defaultdict(lambda: collections.Counter())
inside = False
cur = (None, 0)
for line in open(path, errors="replace"):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', line)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
        continue
    if line.startswith(fn + ":"):
        inside = True
        continue
    if not inside:
        continue
    if line.startswith(".Lfunc_end"):
        break
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', line)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    s = line.strip()
    if not s or s.startswith((".", ";")) or s.endswith(":"):
        continue
    op = s.split()[0]
    kind = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_")
            else "vmem" if op.startswith(("global_", "flat_", "buffer_", "scratch_")) else "other")
    cnt[cur][kind] += 1
tot = collections.Counter()
rows = []
for (f, l), c in cnt.items():
    name = files.get(f, "?")
    tot.update(c)
    if want and want not in name:
        continue
    rows.append((name, l, c))
rows.sort(key=lambda r: (r[0], r[1]))
for name, l, c in rows:
    print("%-20s %5d  valu %4d salu %4d lds %3d vmem %3d" % (name, l, c["valu"], c["salu"], c["lds"], c["vmem"]))
print("TOTAL", dict(tot))
