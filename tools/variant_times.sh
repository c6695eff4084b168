#!/bin/bash
# usage (GPU box): tools/variant_times.sh "<v1> <v2> ..." [kbench args]: kernel time of the iteration kernel per variant lib
# ("default" = the in-tree library), each under rocprofv3 --kernel-trace --stats, all on the same box in one go.
vs=$1; shift
for v in $vs; do
  if [ "$v" = default ]; then unset ICA_LIB; else export ICA_LIB=$GRAFT_REPO_ROOT/tools/abl/lib_$v.so; fi
  timeout -k 10 200 $GRAFT_REPO_ROOT/tools/prof_kbench.sh vt_$v "$@" > /dev/null 2>&1
  echo "$v: $(grep -E 'iterate_kernel' $GRAFT_REPO_ROOT/gpurun_out/vt_$v.txt | awk '{print $1,$2,$3,$4,$5,"calls",$6,"avg_us",$8,"min",$9}' | tr '\n' ';')"
done
