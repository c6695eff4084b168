#!/bin/bash
# usage (GPU box): tools/pmc_variant.sh <variant|default> "<counters>" [kbench args]   -> gpurun_out/pv_<variant>_pmc.txt
v=$1; shift; ctr=$1; shift
if [ "$v" = default ]; then unset ICA_LIB; else export ICA_LIB=$GRAFT_REPO_ROOT/tools/abl/lib_$v.so; fi
$GRAFT_REPO_ROOT/tools/pmc_any.sh pv_$v "$ctr" iterate $GRAFT_REPO_ROOT/tools/kbench.py --reps 3 "$@"
