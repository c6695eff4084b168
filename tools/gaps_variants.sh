#!/bin/bash
# usage (GPU box): tools/gaps_variants.sh "<v1> <v2>" [bench args]: busy / idle time of one solve step per variant lib
vs=$1; shift
cd /tmp && export TMPDIR=/tmp
for v in $vs; do
  if [ "$v" = default ]; then unset ICA_LIB; else export ICA_LIB=$GRAFT_REPO_ROOT/tools/abl/lib_$v.so; fi
  rm -rf /tmp/gv_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/gv_$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-per-config "$@" > $GRAFT_REPO_ROOT/gpurun_out/gv_$v.log 2>&1
  echo "== $v"; python3 $GRAFT_REPO_ROOT/tools/trace_gaps.py /tmp/gv_$v/*/*kernel_trace.csv $GAPS_VERBOSE
done
