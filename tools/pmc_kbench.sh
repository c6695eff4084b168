#!/bin/bash
# usage (on the GPU box): tools/pmc_kbench.sh <tag> "<counters>" [kbench args...]  -> gpurun_out/<tag>_pmc.txt
tag=$1; shift; ctr=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$tag
rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$tag -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --reps 3 "$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc.log 2>&1
python3 - <<PY > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc.txt
import csv, glob, collections
f = glob.glob('/tmp/pmc_$tag/*/*counter_collection.csv')
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name']
        if 'accumulate' not in k and 'iterate' not in k: continue
        agg[k.split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:<28} n={len(v):<3} mean={sum(v)/len(v):.4g}")
PY
cat $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc.txt
