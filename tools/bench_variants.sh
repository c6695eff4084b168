#!/bin/bash
# usage (GPU box): tools/bench_variants.sh "<v1> <v2> ..." [bench args]: ms_per_step of bench.py per variant lib
# ("default" = the in-tree library), same box, one after the other.
vs=$1; shift
for v in $vs; do
  if [ "$v" = default ]; then unset ICA_LIB; else export ICA_LIB=$GRAFT_REPO_ROOT/tools/abl/lib_$v.so; fi
  timeout -k 10 300 python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-per-config "$@" 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v', 'ms_per_step %.3f' % d['ms_per_step'], 'value %.0f' % d['value'], 'frac %.3f' % r['frac'], 'scale0 frac %.3f' % (r['scale0']['frac'] or 0), 'kernel share %.2f' % r['kernel_share_of_step'])"
done
