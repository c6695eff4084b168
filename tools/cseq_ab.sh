#!/bin/bash
# usage (GPU box): tools/cseq_ab.sh -- channel-sequential three-channel tiles (tools/mkvariant.sh cseq -DICA_CSEQ=1) against the
# product's resident-channel kernel.  Results: profiles/r03_ablation.txt, DESIGN.md 5.1.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/cseq_ab; mkdir -p $O
cd $R
echo "== gpu tests default" && timeout -k 10 500 python -m pytest tests -m gpu -x -q > $O/tests_default.log 2>&1; tail -5 $O/tests_default.log
echo "== C=3 robust homography 1080x1920" && tools/variant_times.sh "default cseq" --reps 5 --size 1080 1920 --channels 3 --transform 5 --robust 3 --batch 256 | tee $O/vt_c3_rob.txt
echo "== C=3 L2 homography 1080x1920" && tools/variant_times.sh "default cseq" --reps 5 --size 1080 1920 --channels 3 --transform 5 --robust 0 --batch 256 | tee $O/vt_c3_l2.txt
echo "== C=3 L2 affinity 512" && tools/variant_times.sh "default cseq" --reps 5 --size 512 512 --channels 3 --transform 4 --robust 0 --batch 1024 | tee $O/vt_c3_aff.txt
for v in default cseq; do
  if [ "$v" = default ]; then unset ICA_LIB; else export ICA_LIB=$R/tools/abl/lib_$v.so; fi
  echo "== traffic C=3 $v" && timeout -k 10 300 tools/pmc_traffic.sh cseq_ab_$v --size 1080 1920 --channels 3 --transform 5 --robust 3 --batch 256 | tee $O/traffic_c3_$v.txt
  echo "== bench cfg5 $v" && timeout -k 10 400 python bench.py --config 5 --steps 2 --warmup 1 --no-per-config --no-cpu-baseline > $O/bench_cfg5_$v.json 2> $O/bench_cfg5_$v.err; tail -c 1500 $O/bench_cfg5_$v.json
done
