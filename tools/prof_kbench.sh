#!/bin/bash
# usage (on the GPU box): tools/prof_kbench.sh <tag> [kbench args...]   -> gpurun_out/<tag>.txt
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 $GRAFT_REPO_ROOT/tools/kbench.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/$tag.log 2>&1
python3 $GRAFT_REPO_ROOT/profiles/summarize.py /tmp/prof_$tag/*/*kernel_stats.csv > $GRAFT_REPO_ROOT/gpurun_out/$tag.txt 2>&1
grep -E "wall|checksum" $GRAFT_REPO_ROOT/gpurun_out/$tag.log >> $GRAFT_REPO_ROOT/gpurun_out/$tag.txt
grep -E "iterate|accumulate|wall|checksum" $GRAFT_REPO_ROOT/gpurun_out/$tag.txt
