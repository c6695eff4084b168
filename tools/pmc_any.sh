#!/bin/bash
# usage (GPU box): tools/pmc_any.sh <tag> "<counters>" <kernel-name filter> <python script> [args...] -> gpurun_out/<tag>_pmc.txt
tag=$1; shift; ctr=$1; shift; filt=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmca_$tag
timeout -k 5 150 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmca_$tag -- python3 "$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc.log 2>&1
python3 - "$filt" <<PY > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc.txt
import csv, glob, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob('/tmp/pmca_$tag/*/*counter_collection.csv'):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name']
        if sys.argv[1] not in k: continue
        agg[k.split('(')[0][-48:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:<32} n={len(v):<3} mean={sum(v)/len(v):.5g} max={max(v):.5g}")
PY
cat $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc.txt
