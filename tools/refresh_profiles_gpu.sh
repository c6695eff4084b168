#!/bin/bash
# usage (GPU box): tools/refresh_profiles_gpu.sh <round tag, e.g. r02>
# Every measurement profiles/<tag>_* is assembled from (tools/assemble_profiles.py does that, off the box):
#   bench.py under rocprofv3 --kernel-trace --stats for each benchmark configuration, full-batch launches of each
#   iteration-kernel shape under --kernel-trace --stats and under the two --pmc passes of tools/pmc_traffic.sh,
#   and one plain `python bench.py` (the line the driver records).
t=$1
R=$GRAFT_REPO_ROOT
set -e
# the library these measurements are taken on (bench.py reports roofline.traffic only for this very build)
(cd $R && python3 -c "from inverse_compositional_algorithm_amd import build; print(build.read_stamp()[0])") > $R/gpurun_out/${t}_fingerprint.txt
for cfg in 2 3 4 5; do
  timeout -k 10 400 $R/tools/prof_bench.sh ${t}_bs_cfg$cfg --config $cfg > /dev/null; echo "bench summary cfg $cfg done"
done
timeout -k 10 400 $R/tools/prof_bench.sh ${t}_bs_cfg4_u8 --config 4 --storage u8 > /dev/null; echo "bench summary cfg 4 u8 done"
declare -A shapes=( [h_rob]="--size 1024 1024 --transform 5 --robust 3 --batch 1024" [h_l2]="--size 1024 1024 --transform 5 --robust 0 --batch 1024"
                    [a_l2]="--size 512 512 --transform 4 --robust 0 --batch 1024" [h_rob3]="--size 1080 1920 --channels 3 --transform 5 --robust 3 --batch 256" )
for s in h_rob h_l2 a_l2 h_rob3; do
  timeout -k 10 300 $R/tools/prof_kbench.sh ${t}_kb_$s --reps 5 ${shapes[$s]} > /dev/null; echo "kbench $s done"
  timeout -k 10 400 $R/tools/pmc_traffic.sh ${t}_pmc_$s ${shapes[$s]} > /dev/null; echo "pmc $s done"
done
cd $R && timeout -k 10 900 python3 bench.py > $R/gpurun_out/${t}_bench_line.json 2> $R/gpurun_out/${t}_bench_line.err; echo "bench line done"
tail -c 600 $R/gpurun_out/${t}_bench_line.json
