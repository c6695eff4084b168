#!/bin/bash
# round-3 experiment batch (GPU box): default library's GPU suite, the interleaved walk and the 8-wave C=3 variants
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3f; mkdir -p $O
cd $R
echo "== mfma ubench" && timeout -k 10 120 tools/ubench/mfma_coissue > $O/mfma_coissue.txt 2>&1; cat $O/mfma_coissue.txt
echo "== gpu tests default" && timeout -k 10 400 python -m pytest tests -m gpu -x -q > $O/tests_default.log 2>&1; tail -3 $O/tests_default.log
for v in il2 c3w8il2; do
  echo "== gpu tests $v" && ICA_LIB=$R/tools/abl/lib_$v.so timeout -k 10 400 python -m pytest tests -m gpu -q > $O/tests_$v.log 2>&1; tail -5 $O/tests_$v.log
done
echo "== C=1 robust homography 1024^2" && tools/variant_times.sh "default il2" --reps 5 --size 1024 1024 --transform 5 --robust 3 --batch 1024 | tee $O/vt_c1_rob.txt
echo "== C=1 L2 homography 1024^2" && tools/variant_times.sh "default il2" --reps 5 --size 1024 1024 --transform 5 --robust 0 --batch 1024 | tee $O/vt_c1_l2.txt
echo "== C=1 L2 affinity 512^2" && tools/variant_times.sh "default il2" --reps 5 --size 512 512 --transform 4 --robust 0 --batch 1024 | tee $O/vt_c1_aff.txt
echo "== C=3 robust homography 1080x1920" && tools/variant_times.sh "default il2 c3w8 c3w8il2" --reps 5 --size 1080 1920 --channels 3 --transform 5 --robust 3 --batch 256 | tee $O/vt_c3_rob.txt
for v in default il2 c3w8il2; do
  if [ "$v" = default ]; then unset ICA_LIB; else export ICA_LIB=$R/tools/abl/lib_$v.so; fi
  echo "== traffic C=3 $v" && timeout -k 10 300 tools/pmc_traffic.sh r3f_c3_$v --size 1080 1920 --channels 3 --transform 5 --robust 3 --batch 256 | tee $O/traffic_c3_$v.txt
done
for v in il2; do
  export ICA_LIB=$R/tools/abl/lib_$v.so
  echo "== traffic C=1 $v" && timeout -k 10 300 tools/pmc_traffic.sh r3f_c1_$v --size 1024 1024 --transform 5 --robust 3 --batch 1024 | tee $O/traffic_c1_$v.txt
done
