#!/bin/bash
# Development aid: timeline of the last of a few 10 MiB host-buffer decompress calls (HIP API, kernels, copies).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/host_trace; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
python3 $R/scripts/host_small_call_trace.py 12 2>&1 | grep c_abi
timeout 300 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d "$OUT" -- python3 $R/scripts/host_small_call_trace.py 12 > "$OUT/run.log" 2>&1
grep c_abi "$OUT/run.log"
python3 - "$OUT" <<'PY'
import csv, glob, sys
ev = []
for f in glob.glob(sys.argv[1] + "/**/*_hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "api  " + r["Function"]))
for f in glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "KERN " + r["Kernel_Name"][:40]))
for f in glob.glob(sys.argv[1] + "/**/*_memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Size", "")))
ev.sort()
# the last call: from the last k_decode_cu back to the preceding gap
ks = [e for e in ev if e[2].startswith("KERN k_decode_cu")]
if not ks: sys.exit("no k_decode_cu")
k = ks[-1]
t0 = k[0] - 600000; t1 = k[0] + 700000
for s, e, n in ev:
    if s >= t0 and s <= t1: print("%9.1f us  +%8.1f us  %s" % ((s - k[0]) / 1e3, (e - s) / 1e3, n))
PY
