#!/bin/bash
# usage: tools/pmc_any.sh <tag> "<counters>" <kernel substring> <script> [args]
tag=$1; shift; ctr=$1; shift; pat=$1; shift
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pmc_$tag
timeout 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_$tag -- python3 $GRAFT_REPO_ROOT/"$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc.log 2>&1
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in glob.glob('/tmp/pmc_$tag/*/*counter_collection.csv'):
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name']
        if '$pat' not in k: continue
        agg[k.split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
dur = collections.defaultdict(list)
for fn in glob.glob('/tmp/pmc_$tag/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(fn)):
        if '$pat' in r['Kernel_Name']:
            dur[r['Kernel_Name'].split('(')[0][-40:]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, d in agg.items():
    print(k, "avg_us=%.1f" % (sum(dur[k]) / max(1, len(dur[k]))) if k in dur else "")
    for c, v in sorted(d.items()):
        print(f"   {c:<28} n={len(v):<3} max={max(v):.4g} mean={sum(v)/len(v):.4g}")
PY
