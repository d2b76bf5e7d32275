#!/bin/bash
# Development aid: kernel timeline of ONE reference-written linked text stream through the run-in decode (rocprofv3 --kernel-trace
# --stats; the summary goes to gpurun_out/runin_prof/kernel_stats_<blocks>.csv):   bash scripts/prof_runin.sh [blocks=16384]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NB=${1:-16384}
OUT=$R/gpurun_out/runin_prof
mkdir -p "$OUT"; rm -rf "$OUT/run_$NB"
cd /tmp && export TMPDIR=/tmp
RUNIN_ONLY=default rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/run_$NB" -- python3 "$R/scripts/linked_runin_time.py" text $NB > "$OUT/run_$NB.log" 2>&1
cp "$OUT"/run_$NB/*/*kernel_stats.csv "$OUT/kernel_stats_$NB.csv" 2>/dev/null
grep -i "runin\|decode_par\|link_stat\|longest" "$OUT/kernel_stats_$NB.csv" | cut -d, -f1-4,6,7
