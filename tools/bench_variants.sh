#!/bin/bash
# usage (GPU box): tools/bench_variants.sh "<v1> <v2> ..." [bench args]: ms_per_step of bench.py per variant lib (2 runs each)
vs=$1; shift
for v in $vs; do
  for r in 1 2; do
    ICA_LIB=$GRAFT_REPO_ROOT/tools/abl/lib_$v.so timeout 200 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'ms_per_step %.3f' % d['ms_per_step'], 'avg_launch_ms %.4f' % d['roofline']['avg_launch_ms'])"
  done
done
