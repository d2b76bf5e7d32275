#!/bin/bash
# Development aid (round 6, review item 4): the encoder with a 3072-entry table (7.5 KiB of LDS: 20 waves per CU) at four and at five
# waves per SIMD against the shipped one (4096 entries, 10 KiB, four waves): rate, ratio, registers, and the SQ wait / busy counters.
# Build the variants first (CPU side):
#   scripts/ab_variants.sh tab3072w4 "-DENC_TAB_N=3072" tab3072w5 "-DENC_TAB_N=3072 -DENC_WAVES_PER_EU=5"
# then on the GPU box:  bash scripts/enc_occupancy_experiment.sh > gpurun_out/encode_occupancy_experiment.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in default tab3072w4 tab3072w5; do
  if [ "$v" = default ]; then unset MI355LZ4_LIB; LIBF=$R/streamly-lz4_amd/lib/libmi355lz4.so; else export MI355LZ4_LIB=$R/streamly-lz4_amd/lib/variants/$v.so; LIBF=$MI355LZ4_LIB; fi
  echo "== $v"
  bash $R/scripts/kernel_resources.sh $LIBF | grep "k_encodeILb0ELb1"
  python3 $R/scripts/enc_quick.py lzsynth,text 32768 2>&1 | grep "GB/s"
  for kind in lzsynth text; do
    OUT=$R/gpurun_out/pmc_occ_${v}_$kind; rm -rf "$OUT"; mkdir -p "$OUT"
    i=0
    for grp in "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_LDS"; do
      i=$((i+1))
      rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -- python3 "$R/scripts/prof_encode.py" $kind 32768 2 > "$OUT/g$i.log" 2>&1
    done
    python3 - "$OUT" "$kind" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/g*/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_encode" in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print("  %-8s " % sys.argv[2] + "  ".join("%s %.4g" % (k, m[k]) for k in sorted(m)))
if "SQ_WAIT_ANY" in m and "SQ_WAVE_CYCLES" in m:
    print("  %-8s waiting %.1f %% of wave cycles; wave cycles per busy cycle (resident waves per SIMD, all SIMDs) %.2f" % (sys.argv[2], 100 * m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], m["SQ_WAVE_CYCLES"] / m["SQ_BUSY_CYCLES"] if m.get("SQ_BUSY_CYCLES") else 0))
PY
  done
done
