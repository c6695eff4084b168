#!/bin/bash
# usage (GPU box): tools/ablate_pipes.sh -- which pipe is the iteration kernel sensitive to?  Timing-only builds, made first with
#   for v in 16 64 2 18 66 82 1; do tools/mkvariant.sh a$v -DICA_ABLATE=$v; done
# (ICA_ABLATE bits: ica_iterate.hpp).  Results: profiles/r03_ablation.txt, DESIGN.md 5.1.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ablate_pipes; mkdir -p $O
cd $R
echo "== C=1 robust homography 1024^2 x 1024" && tools/variant_times.sh "default a16 a64 a2 a18 a66 a82 a1" --reps 5 --size 1024 1024 --transform 5 --robust 3 --batch 1024 | tee $O/vt_c1_rob.txt
echo "== C=3 robust homography" && tools/variant_times.sh "default a16 a64 a2 a18 a1" --reps 5 --size 1080 1920 --channels 3 --transform 5 --robust 3 --batch 256 | tee $O/vt_c3_rob.txt
echo "== C=1 L2 affinity 512" && tools/variant_times.sh "default a16 a2 a18 a1" --reps 5 --size 512 512 --transform 4 --robust 0 --batch 1024 | tee $O/vt_c1_aff.txt
