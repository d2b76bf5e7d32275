# Development aid: kernel times of the linked second pass for the main library and every variant under lib/variants/
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for lib in $R/streamly-lz4_amd/lib/libmi355lz4.so $R/streamly-lz4_amd/lib/variants/*.so; do
  [ -f "$lib" ] || continue
  n=$(basename $lib .so)
  echo "== $n"
  export MI355LZ4_LIB=$lib
  rm -rf $R/gpurun_out/loc_prof_$n
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/loc_prof_$n -o loc -- python3 $R/scripts/linked_local_ab.py ${1:-4096} ${2:-text} 1 2>&1 | grep blocks
  python3 $R/scripts/rocpd_kernels.py $R/gpurun_out/loc_prof_$n/loc_results.db 2>&1 | grep -E "k_loc|k_decode_tol"
  rm -rf $R/gpurun_out/loc_prof_$n
done
