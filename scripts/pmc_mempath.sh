#!/bin/bash
# Development aid (round 6): the vector-memory path (TA / TCP / TD) counters of the encode and decode kernels: is the texture addresser,
# not instruction issue, what the scattered 16-byte loads wait for?   usage: pmc_mempath.sh enc|dec [kind]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
WHAT=${1:-enc}; KIND=${2:-lzsynth}
OUT=$R/gpurun_out/pmc_mem_${WHAT}_${KIND}${TAG:+_$TAG}
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
if [ "$WHAT" = enc ]; then PROG="$R/scripts/prof_encode.py $KIND 32768 2"; KER=k_encode; else PROG="$R/scripts/prof_decode.py $KIND 32768 2"; KER=k_decode_par; fi
i=0
# (TA_ADDR_STALLED_* / TCP_* groups made rocprofv3 abort and hang on this pool: two groups that are known to work, each under a timeout)
for grp in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE TA_TOTAL_WAVEFRONTS_sum" \
           "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -- python3 $PROG > "$OUT/g$i.log" 2>&1
done
python3 - "$OUT" $KER "$WHAT $KIND" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if sys.argv[2] in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("# %s: per launch of %s over 32 768 blocks of 64 KiB (the largest launches only)" % (sys.argv[3], sys.argv[2]))
for k in sorted(acc):
    v = sorted(acc[k])[-2:]
    print("%-40s %.4g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
