#!/bin/bash
# Development aid: instruction-mix / busy counters of the workgroup-per-block decoder (one --pmc pass per group).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
KIND=${1:-lzsynth}
OUT=$R/gpurun_out/pmc_cu_$KIND
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -- python3 "$R/scripts/prof_cu.py" $KIND 160 4 > "$OUT/g$i.log" 2>&1
done
python3 - "$OUT" k_decode_cu <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if sys.argv[2] in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
print("# per launch of k_decode_cu: 160 blocks of 64 KiB (10 MiB), one workgroup of 1024 threads per block")
for k in sorted(acc):
    v = acc[k]
    print("%-28s %.4g  (n=%d)" % (k, sum(v) / len(v), len(v)))
PY
