#!/bin/bash
# usage: tools/mkvariant.sh <name> <flags...>  -> tools/abl/lib_<name>.so (select with ICA_LIB=tools/abl/lib_<name>.so)
name=$1; shift
mkdir -p "$(dirname "$0")/abl"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o "$(dirname "$0")/abl/lib_$name.so" \
  "$(dirname "$0")/../inverse_compositional_algorithm_amd/csrc/ica_api.hip" "$@" 2>&1 | grep -E "error|static assertion" 
