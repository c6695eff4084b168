#!/bin/bash
# usage (GPU box): tools/launch_profile.sh <tag> [bench args]: per-launch duration and grid of the iteration kernels of ONE solve
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/lp_$tag
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/lp_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/lp_$tag.log 2>&1
python3 - <<PY > $GRAFT_REPO_ROOT/gpurun_out/lp_$tag.txt
import csv, glob
rows = []
for fn in glob.glob('/tmp/lp_$tag/*/*kernel_trace.csv'):
    rows += list(csv.DictReader(open(fn)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
import os
acc = [r for r in rows if os.environ.get('LP_KERNEL', 'accumulate2') in r['Kernel_Name']]
half = acc[len(acc) // 2:]   # the timed solve (second of two)
tot = 0
for r in half:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    print(r['Kernel_Name'].split('(')[0][-28:], 'grid', int(r['Grid_Size_X']) // 256 if 'Grid_Size_X' in r else r.get('Grid_Size'), 'us %.1f' % d)
print('total_us %.1f' % tot)
PY
tail -45 $GRAFT_REPO_ROOT/gpurun_out/lp_$tag.txt
