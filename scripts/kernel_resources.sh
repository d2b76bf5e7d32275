#!/bin/bash
# Development aid: per-kernel register / LDS / spill figures of the built library's gfx950 code object.
set -e
cd "$(dirname "$0")/.."
LIB=${1:-streamly-lz4_amd/lib/libmi355lz4.so}
TMP=$(mktemp -d)
/opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$LIB" --output="$TMP/dev.co" --unbundle 2>/dev/null || \
  /opt/rocm/bin/roc-obj-ls "$LIB" >/dev/null 2>&1 || true
if [ ! -s "$TMP/dev.co" ]; then
  # shared library: the fat binary sits in .hip_fatbin
  /opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin "$LIB" "$TMP/fat.bin"
  /opt/rocm/lib/llvm/bin/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input="$TMP/fat.bin" --output="$TMP/dev.co" --unbundle
fi
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$TMP/dev.co" | python3 -c '
import sys, re
cur = {}
for line in sys.stdin:
    m = re.match(r"\s*-?\s*\.?(\w+):\s*(.*)", line)
    if not m: continue
    k, v = m.group(1), m.group(2).strip()
    if k == "name" and v.startswith("_Z") or k == "name" and v.startswith("k_"):
        cur["name"] = v
    if k in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "group_segment_fixed_size", "private_segment_fixed_size"):
        cur[k] = v
    if k == "wavefront_size" and "name" in cur:
        print("%-60s vgpr %3s sgpr %3s vspill %2s sspill %3s lds %6s scratch %4s" % (cur.get("name","?")[:60], cur.get("vgpr_count"), cur.get("sgpr_count"), cur.get("vgpr_spill_count"), cur.get("sgpr_spill_count"), cur.get("group_segment_fixed_size"), cur.get("private_segment_fixed_size")))
        cur = {}
'
rm -rf "$TMP"
