#!/bin/bash
# usage (GPU box): tools/prof_variants.sh "<v1> <v2> ..." [kbench args]  -> kernel avg durations per variant lib in tools/abl/
vs=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in $vs; do
  rm -rf /tmp/pv_$v
  ICA_LIB=$GRAFT_REPO_ROOT/tools/abl/lib_$v.so timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv_$v -- python3 $GRAFT_REPO_ROOT/tools/kbench.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/pv_$v.log 2>&1 || { echo "$v FAILED"; tail -3 $GRAFT_REPO_ROOT/gpurun_out/pv_$v.log; exit 1; }
  echo "== $v: $(python3 $GRAFT_REPO_ROOT/profiles/summarize.py /tmp/pv_$v/*/*kernel_stats.csv | grep accumulate | awk '{printf "%s %s avg_us=%s | ", $2,$3,$7}') $(grep checksum $GRAFT_REPO_ROOT/gpurun_out/pv_$v.log)"
done
