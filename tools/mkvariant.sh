#!/bin/bash
# usage: tools/mkvariant.sh <name> <flags...>  -> tools/abl/lib_<name>.so (select with ICA_LIB=tools/abl/lib_<name>.so)
name=$1; shift
root="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$root/tools/abl"
cd "$root" && python3 -c "
import sys
from inverse_compositional_algorithm_amd import build
print(build.build_variant('$root/tools/abl/lib_$name.so', sys.argv[1:]))" "$@"
