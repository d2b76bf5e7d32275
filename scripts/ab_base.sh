#!/bin/bash
# Development aid: build the library of a given commit (default HEAD) into streamly-lz4_amd/lib/variants/<name>.so so that
# scripts/ab_time.py can time it beside the working tree's build.   usage: scripts/ab_base.sh [commit] [name] [extra flags]
set -e
cd "$(dirname "$0")/.."
C=${1:-HEAD}; N=${2:-base}; F=${3:-}
T=build/ab_src_$N
rm -rf "$T"; mkdir -p "$T" streamly-lz4_amd/lib/variants
git archive "$C" streamly-lz4_amd/csrc include | tar -x -C "$T"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result $F -shared -Wl,-Bsymbolic \
   -o streamly-lz4_amd/lib/variants/$N.so -x hip "$T"/streamly-lz4_amd/csrc/kernels.hip "$T"/streamly-lz4_amd/csrc/api.cpp \
   "$T"/streamly-lz4_amd/csrc/host_stream.cpp "$T"/streamly-lz4_amd/csrc/lz4_frame.cpp 2>&1 | grep -E "error" || true
echo "built $N from $C"
