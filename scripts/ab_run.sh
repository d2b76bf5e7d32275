#!/bin/bash
# Development aid: run scripts/ab_dec.py against the main library and every variant under lib/variants/.
#   scripts/ab_run.sh [kinds] [nblocks] [decoder variants]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for lib in $R/streamly-lz4_amd/lib/libmi355lz4.so $R/streamly-lz4_amd/lib/variants/*.so; do
  [ -f "$lib" ] || continue
  echo "== $(basename $lib)"
  MI355LZ4_LIB=$lib python3 $R/scripts/ab_dec.py "${1:-lzsynth,text}" "${2:-32768}" "${3:-2}" 2>&1 | grep -v amdgpu.ids
done
