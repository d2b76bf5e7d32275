"""Development aid: per-kernel dispatch statistics from a rocprofv3 rocpd (.db) result.
    python scripts/rocpd_kernels.py <results.db>"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else cols[0]
rows = cur.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
agg = collections.OrderedDict()
for n, s, e in rows:
    a = agg.setdefault(n, []); a.append((e - s) / 1e6)
for n, v in agg.items():
    v2 = sorted(v)
    print("%-70s n=%3d  min %.3f  med %.3f  avg %.3f ms" % (n[:70], len(v), v2[0], v2[len(v2) // 2], sum(v) / len(v)))
