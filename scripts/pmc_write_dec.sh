#!/bin/bash
# Development aid: WRITE_SIZE / FETCH_SIZE of the decode kernel for one library build (MI355LZ4_LIB), lzsynth 32768 blocks.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_wr_${1:-main}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in WRITE_SIZE FETCH_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d "$OUT/$c" -- python3 "$R/scripts/prof_decode.py" ${2:-lzsynth} 32768 2 > "$OUT/$c.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_decode_par" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print("%-12s %.5g KB per launch (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
