#!/bin/bash
# round-3 batch 3 (GPU box): which pipe is the iteration kernel sensitive to?  timing-only builds
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3h; mkdir -p $O
cd $R
echo "== C=1 robust homography 1024^2 x 1024" && tools/variant_times.sh "default a16 a64 a2 a18 a66 a82 a1" --reps 5 --size 1024 1024 --transform 5 --robust 3 --batch 1024 | tee $O/vt_c1_rob.txt
echo "== C=3 robust homography" && tools/variant_times.sh "default a16 a64 a2 a18 a1" --reps 5 --size 1080 1920 --channels 3 --transform 5 --robust 3 --batch 256 | tee $O/vt_c3_rob.txt
echo "== C=1 L2 affinity 512" && tools/variant_times.sh "default a16 a2 a18 a1" --reps 5 --size 512 512 --transform 4 --robust 0 --batch 1024 | tee $O/vt_c1_aff.txt
