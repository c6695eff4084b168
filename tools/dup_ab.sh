#!/bin/bash
# usage (GPU box): tools/dup_ab.sh -- one-channel float32 tiles with a second, one-float-shifted copy of the I2 footprint (aligned
# 8-byte tap reads; tools/mkvariant.sh dup -DICA_DUP=1) against the product kernel.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/dup_ab; mkdir -p $O
cd $R
echo "== gpu tests with the dup library (uint8-vs-float32 bit-identity tests are expected to fail: other chunk height)"
ICA_LIB=$R/tools/abl/lib_dup.so timeout -k 10 600 python -m pytest tests -m gpu -q > $O/tests_dup.log 2>&1; tail -15 $O/tests_dup.log | cut -c1-160
echo "== robust homography 1024^2 x 1024" && tools/variant_times.sh "default dup" --reps 5 --size 1024 1024 --transform 5 --robust 3 --batch 1024 | tee $O/vt_rob.txt
echo "== L2 homography" && tools/variant_times.sh "default dup" --reps 5 --size 1024 1024 --transform 5 --robust 0 --batch 1024 | tee $O/vt_l2.txt
echo "== L2 affinity 512^2" && tools/variant_times.sh "default dup" --reps 5 --size 512 512 --transform 4 --robust 0 --batch 1024 | tee $O/vt_aff.txt
for v in default dup; do
  if [ "$v" = default ]; then unset ICA_LIB; else export ICA_LIB=$R/tools/abl/lib_$v.so; fi
  for c in "4" "3" "2 --in-flight 3"; do
    echo "$v bench --config $c: $(timeout -k 10 400 python bench.py --config $c --no-per-config --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step %.3f value %.0f frac %.3f' % (d['ms_per_step'], d['value'], d['roofline']['frac']))")"
  done
done | tee $O/bench_ab.txt
unset ICA_LIB
export ICA_LIB=$R/tools/abl/lib_dup.so
echo "== traffic dup" && timeout -k 10 300 tools/pmc_traffic.sh dup_c1 --size 1024 1024 --transform 5 --robust 3 --batch 1024 | tee $O/traffic_dup.txt
