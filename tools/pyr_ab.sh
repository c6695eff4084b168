#!/bin/bash
# usage (GPU box): tools/pyr_ab.sh <label> -- pyramid kernel times (rocprofv3 --kernel-trace --stats, tools/pyrbench.py --batch 1024
# --reps 5) at four level sizes + the benchmark configurations the pyramid weighs most in, for the library in the tree.
set -o pipefail
L=${1:-cur}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pyr_ab; mkdir -p $O
cd $R
echo "== gpu tests" && timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/tests_$L.log 2>&1; tail -3 $O/tests_$L.log
cd /tmp && export TMPDIR=/tmp
for sz in "1024 1024" "512 512" "256 256" "1080 1920"; do
  rm -rf /tmp/prof_pyr
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pyr -- python3 $R/tools/pyrbench.py --batch 1024 --reps 5 --size $sz > $O/prof.log 2>&1
  echo "$L size $sz: $(grep wall $O/prof.log) | $(python3 $R/profiles/summarize.py /tmp/prof_pyr/*/*kernel_stats.csv | grep -E 'resample' | awk '{print $1, "calls", $(NF-5), "avg_us", $(NF-3)}' | tr '\n' ';')"
done | tee $O/pyr_kernel_times_$L.txt
cd $R
for c in "2 --in-flight 3" "4" "5 --steps 3 --warmup 1" "4 --storage u8"; do
  echo "$L bench --config $c: $(timeout -k 10 400 python bench.py --config $c --no-per-config --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step %.3f value %.0f frac %.3f' % (d['ms_per_step'], d['value'], d['roofline']['frac']))")"
done | tee $O/bench_$L.txt
