#!/bin/bash
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3i; mkdir -p $O
cd $R
echo "== gpu tests" && timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
echo "== smoke" && timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== bench" && timeout -k 10 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json; tail -3 $O/bench.err
