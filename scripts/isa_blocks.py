#!/usr/bin/env python3
"""Development aid: basic blocks of one kernel's ISA with instruction counts per kind and branch targets.
usage: isa_blocks.py kernel.s   (the function's text cut out of hipcc -S output)"""
import re, sys, collections
blocks = []; cur = None
for line in open(sys.argv[1]):
    s = line.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', s)
    if m:
        cur = [m.group(1), collections.Counter(), [], 0]; blocks.append(cur); continue
    if cur is None:
        cur = ['entry', collections.Counter(), [], 0]; blocks.append(cur)
    if not s or s.startswith(('.', ';')) or s.endswith(':'): continue
    op = s.split()[0]
    kind = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_")
            else "vmem" if op.startswith(("global_", "flat_", "buffer_", "scratch_")) else "other")
    cur[1][kind] += 1
    if op.startswith('s_cbranch') or op == 's_branch':
        cur[2].append(op[2:] + '->' + s.split()[-1])
for b in blocks:
    c = b[1]
    print("%-12s v%4d s%4d l%3d m%3d  %s" % (b[0], c['valu'], c['salu'], c['lds'], c['vmem'], ' '.join(b[2])))
