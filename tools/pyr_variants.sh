#!/bin/bash
# usage (GPU box): tools/pyr_variants.sh "<v1> <v2> ..." [bench args]: pyramid kernel times of bench.py per variant lib
vs=$1; shift
for v in $vs; do
  if [ "$v" = default ]; then unset ICA_LIB; else export ICA_LIB=$GRAFT_REPO_ROOT/tools/abl/lib_$v.so; fi
  timeout -k 10 300 $GRAFT_REPO_ROOT/tools/prof_bench.sh pv_$v "$@" > /dev/null 2>&1
  echo "== $v: $(python3 -c "import json;d=json.loads(open('$GRAFT_REPO_ROOT/gpurun_out/pv_${v}_line.json').read().strip().splitlines()[-1]);print('ms_per_step %.3f'%d['ms_per_step'])")"
  grep -E "resample|minmax" $GRAFT_REPO_ROOT/gpurun_out/pv_${v}_summary.txt | awk '{printf "   %-52s calls %s total_ms %s avg %s max %s\n",$1,$(NF-5),$(NF-4),$(NF-3),$(NF-1)}'
done
