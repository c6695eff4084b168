#!/bin/bash
# usage: tools/pmc_variant.sh <variant> "<counters>" [kbench args]
v=$1; shift; ctr=$1; shift
export ICA_LIB=$GRAFT_REPO_ROOT/tools/abl/lib_$v.so
exec $GRAFT_REPO_ROOT/tools/pmc_any.sh $v "$ctr" accumulate2 tools/kbench.py --reps 3 "$@"
