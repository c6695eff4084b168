#!/bin/bash
# usage (GPU box): tools/aux_ab.sh -- cache policy of the iteration kernel's tile copies (ICA_DMA_AUX1 / ICA_DMA_AUX2: 2 = nt, 1 = sc0)
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/aux_ab; mkdir -p $O
cd $R
V="default sc12 sc1x sc01 sc0i1 sc0i2 default sc12"
echo "== robust homography 1024^2 x 1024" && tools/variant_times.sh "$V" --reps 5 --size 1024 1024 --transform 5 --robust 3 --batch 1024 | tee $O/vt_rob.txt
echo "== L2 affinity 512^2" && tools/variant_times.sh "$V" --reps 5 --size 512 512 --transform 4 --robust 0 --batch 1024 | tee $O/vt_aff.txt
echo "== C=3 robust" && tools/variant_times.sh "$V" --reps 5 --size 1080 1920 --channels 3 --transform 5 --robust 3 --batch 256 | tee $O/vt_c3.txt
