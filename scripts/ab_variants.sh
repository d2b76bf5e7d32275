#!/bin/bash
# Development aid: build variants of the library with different -D settings into
# streamly-lz4_amd/lib/variants/<name>.so   usage: scripts/ab_variants.sh name "-DPAR_RING=6144 ..." [name flags]...
set -e
cd "$(dirname "$0")/.."
mkdir -p streamly-lz4_amd/lib/variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result $flags -shared -Wl,-Bsymbolic \
     -o streamly-lz4_amd/lib/variants/$name.so -x hip streamly-lz4_amd/csrc/kernels.hip streamly-lz4_amd/csrc/api.cpp streamly-lz4_amd/csrc/host_stream.cpp streamly-lz4_amd/csrc/lz4_frame.cpp 2>&1 | grep -E "error|occupancy" || true
  echo "built $name ($flags)"
done
