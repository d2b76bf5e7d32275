#!/bin/bash
# Development aid: HBM-side traffic of the encode kernel (FETCH_SIZE, WRITE_SIZE: one --pmc pass each) for one library build.
#   scripts/enc_traffic.sh <lib.so> <kind> [nblocks]   -> prints bytes per launch and the ratio to the algorithmic bytes
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
LIB=$1; KIND=${2:-lzsynth}; NB=${3:-32768}
OUT=$R/gpurun_out/enc_traffic_$(basename $LIB .so)_$KIND
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export MI355LZ4_LIB=$LIB
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$OUT/$c" -- python3 "$R/scripts/prof_encode.py" $KIND $NB 2 > "$OUT/$c.log" 2>&1
done
python3 - "$OUT" "$KIND" "$NB" <<'PY'
import csv, glob, sys, re
out, kind, nb = sys.argv[1], sys.argv[2], int(sys.argv[3])
v = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    vals = []
    for f in glob.glob(out + "/" + c + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_encode" in row["Kernel_Name"] and row["Counter_Name"] == c:
                vals.append(float(row["Counter_Value"]))
    v[c] = sum(vals) / max(len(vals), 1)
C = None
for line in open(out + "/FETCH_SIZE.log"):
    m = re.match(r"C (\d+)", line)
    if m: C = int(m.group(1))
U = nb * 65536
traffic = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
print("%s %s: FETCH_SIZE %.3e KiB, WRITE_SIZE %.3e KiB -> %.2f GB per launch; algorithmic U + C = %.2f GB; ratio %.2f" % (out.split("/")[-1], kind, v["FETCH_SIZE"], v["WRITE_SIZE"], traffic / 1e9, (U + (C or 0)) / 1e9, traffic / (U + (C or 1))))
PY
