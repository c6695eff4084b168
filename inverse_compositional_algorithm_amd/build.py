"""Build libica_hip.so (the C-ABI library, include/ica_hip.h) for gfx950 with hipcc, in-tree.

The rebuild decision does not trust timestamps: a stamp file next to the library holds a hash of every source the
library is compiled from (csrc/*.hip, csrc/*.hpp, include/*.h) and of the compile command.  Any edit to any of them,
or a change of flags (ICA_EXTRA_FLAGS), makes the library stale."""
from __future__ import annotations

import glob
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
SRC = os.path.join(CSRC, "ica_api.hip")
LIB = os.path.join(HERE, "libica_hip.so")
STAMP = LIB + ".stamp"


def deps() -> list:
    """Every file the library is built from."""
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp"))
                  + glob.glob(os.path.join(INCLUDE, "*.h")))


DEPS = deps()


def command() -> list:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", LIB, SRC]
    cmd += os.environ.get("ICA_EXTRA_FLAGS", "").split()  # e.g. -DICA_ABLATE=1 for timing-only experiments
    return cmd


def fingerprint() -> str:
    h = hashlib.sha256()
    h.update(" ".join(command()[1:]).encode())
    for d in deps():
        h.update(os.path.relpath(d, HERE).encode())
        with open(d, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def lib_path() -> str:
    return LIB


def is_stale() -> bool:
    if not os.path.isfile(LIB) or not os.path.isfile(STAMP):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != fingerprint()


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    cmd = command()
    if verbose:
        print(" ".join(cmd))
    fp = fingerprint()
    subprocess.run(cmd, check=True, cwd=HERE)
    with open(STAMP, "w") as fh:
        fh.write(fp + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
