#!/bin/bash
# usage (GPU box): tools/pyr_pair_ab.sh -- pyramid step with the adjacent-output horizontal pass (default; pw4 = the same at 4
# waves per SIMD, tools/mkvariant.sh pw4 -DICA_PAIR_WAVES=4) against the generic one (ICA_NO_PAIR_PASS=1): kernel time under
# rocprofv3 --kernel-trace --stats at four level sizes, then the two benchmark configurations the pyramid weighs most in.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pyr_pair; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in pair generic pw4; do
  unset ICA_NO_PAIR_PASS ICA_LIB
  [ $v = generic ] && export ICA_NO_PAIR_PASS=1
  [ $v = pw4 ] && export ICA_LIB=$R/tools/abl/lib_pw4.so
  for sz in "1024 1024" "512 512" "256 256" "1080 1920"; do
    rm -rf /tmp/prof_pyr
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_pyr -- python3 $R/tools/pyrbench.py --batch 1024 --reps 5 --size $sz > $O/prof.log 2>&1
    echo "$v size $sz: $(grep wall $O/prof.log) | $(python3 $R/profiles/summarize.py /tmp/prof_pyr/*/*kernel_stats.csv | grep -E 'resample' | awk '{print $1, "avg_us", $4}' | tr '\n' ';')"
  done
done | tee $O/pyr_kernel_times.txt
cd $R
for v in pair generic; do
  unset ICA_NO_PAIR_PASS; [ $v = generic ] && export ICA_NO_PAIR_PASS=1
  for c in "2 --in-flight 3" "4" "5 --steps 3 --warmup 1"; do
    echo "$v bench --config $c: $(timeout -k 10 400 python bench.py --config $c --no-per-config --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step %.3f value %.0f frac %.3f' % (d['ms_per_step'], d['value'], d['roofline']['frac']))")"
  done
done | tee $O/bench_ab.txt
