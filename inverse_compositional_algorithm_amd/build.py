"""Build libica_hip.so (the C-ABI library, include/ica_hip.h) for gfx950 with hipcc, in-tree."""
from __future__ import annotations

import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "ica_api.hip")
DEPS = [SRC] + [os.path.join(HERE, "csrc", f) for f in ("ica_kernels.hpp", "ica_math.hpp", "ica_pyramid.hpp")] + [
    os.path.join(os.path.dirname(HERE), "include", "ica_hip.h")]
LIB = os.path.join(HERE, "libica_hip.so")


def lib_path() -> str:
    return LIB


def is_stale() -> bool:
    if not os.path.isfile(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS if os.path.isfile(d))


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", LIB, SRC]
    cmd += os.environ.get("ICA_EXTRA_FLAGS", "").split()  # e.g. -DICA_ABLATE=1 for timing-only experiments
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True, cwd=HERE)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
