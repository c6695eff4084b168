#!/bin/bash
# usage (GPU box): tools/prof_bench.sh <tag> [bench.py args]  -> gpurun_out/<tag>_summary.txt (+ the bench line)
# rocprofv3 --kernel-trace --stats over bench.py itself; summary of this library's kernels.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pb_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pb_$tag -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-per-config "$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_line.json 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}.err
python3 $GRAFT_REPO_ROOT/profiles/summarize.py /tmp/pb_$tag/*/*kernel_stats.csv > $GRAFT_REPO_ROOT/gpurun_out/${tag}_summary.txt 2>&1
head -8 $GRAFT_REPO_ROOT/gpurun_out/${tag}_summary.txt
