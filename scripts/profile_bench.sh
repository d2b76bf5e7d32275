#!/bin/bash
# Collect the rocprofv3 evidence profiles/ holds for one bench.py workload.
#   scripts/profile_bench.sh <tag> [bench.py args...]
# Writes under gpurun_out/prof_<tag>/ : stats/ (kernel trace + stats), fetch/, write/ (one PMC pass
# each, no tracing domains mixed in) and bench.json (bench.py's own line from an unprofiled run).
# scripts/summarize_profiles.py then copies the summaries into profiles/.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
echo "${PROFILE_COMMIT:-}" > "$OUT/commit.txt"
cd /tmp && export TMPDIR=/tmp
python3 "$R/bench.py" --steps 10 --warmup 2 "$@" > "$OUT/bench.json" 2> "$OUT/bench.err"
# (--no-extra: every launch of the timed kernel in this run is the workload's full-size launch -- setup's verification
# pass, the warm-up and the timed steps -- so that the kernel's AverageNs in kernel_stats.csv IS roofline.avg_launch_ms)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- \
    python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --no-host-api --no-extra "$@" > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.err"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- \
    python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-host-api --no-extra "$@" > /dev/null 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- \
    python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-host-api --no-extra "$@" > /dev/null 2> "$OUT/write.err"
cat "$OUT/bench.json"
