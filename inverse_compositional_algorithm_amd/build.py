"""Build libica_hip.so (the C-ABI library, include/ica_hip.h) for gfx950 with hipcc, in-tree.

Translation units: csrc/ica_api.hip (C ABI, host driver, the small kernels) and csrc/ica_iterate_inst.hip compiled once
per transform type (-DICA_INST_TT=1..5: the instantiations of the fused iteration kernel).  The six objects are
compiled in parallel and linked into one shared library.

The rebuild decision does not trust timestamps: a stamp file next to the library holds a hash of every source the
library is compiled from (csrc/*.hip, csrc/*.hpp, include/*.h) and of the compile commands.  Any edit to any of them,
or a change of flags (ICA_EXTRA_FLAGS), makes the library stale."""
from __future__ import annotations

import glob
import hashlib
import os
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
SRC = os.path.join(CSRC, "ica_api.hip")
LIB = os.path.join(HERE, "libica_hip.so")
STAMP = LIB + ".stamp"
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
# The iteration kernel's units only.  Packed fp32 (v_pk_*) buys nothing on gfx950 when every operand is a VGPR (one packed
# instruction issues in the time of two plain ones) and the register shuffling it needs costs issue slots (DESIGN.md
# section 5).  The pyramid kernels in ica_api.hip keep it: their multiply-adds take a wave-uniform weight from an SGPR,
# a form that costs 2.1 ns plain and 2.06 ns packed -- two for the price of one.
NO_SLP = ["-fno-slp-vectorize"]


def deps() -> list:
    """Every file the library is built from."""
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp"))
                  + glob.glob(os.path.join(INCLUDE, "*.h")))


DEPS = deps()


def units() -> list:
    """(source, extra flags, object name) of every translation unit."""
    inst = os.path.join(CSRC, "ica_iterate_inst.hip")
    return [(os.path.join(CSRC, "ica_api.hip"), [], "ica_api.o")] + \
           [(inst, NO_SLP + [f"-DICA_INST_TT={t}"], f"ica_iterate_tt{t}.o") for t in range(1, 6)]


def hipcc() -> str:
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def extra_flags() -> list:
    return os.environ.get("ICA_EXTRA_FLAGS", "").split()  # e.g. -DICA_ABLATE=1 for timing-only experiments


def commands(objdir: str, out: str = None, extra: list = None) -> list:
    more = extra_flags() if extra is None else list(extra)
    cmds = [[hipcc()] + CFLAGS + flags + more + ["-c", src, "-o", os.path.join(objdir, obj)]
            for src, flags, obj in units()]
    link = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out or LIB] + \
           [os.path.join(objdir, o) for _, _, o in units()]
    return cmds + [link]


def fingerprint() -> str:
    h = hashlib.sha256()
    for cmd in commands("<objdir>"):
        h.update(" ".join(os.path.relpath(c, HERE) if os.path.isabs(c) and c.startswith(os.path.dirname(HERE)) else c
                          for c in cmd[1:]).encode())
    for d in deps():
        h.update(os.path.relpath(d, HERE).encode())
        with open(d, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def lib_path() -> str:
    return LIB


def lib_digest(path: str = None) -> str:
    """sha256 of the shared library's bytes."""
    h = hashlib.sha256()
    with open(path or LIB, "rb") as fh:
        for block in iter(lambda: fh.read(1 << 20), b""):
            h.update(block)
    return h.hexdigest()


def read_stamp() -> tuple:
    """(source fingerprint, library digest) recorded by the last build, ('', '') without a stamp."""
    if not os.path.isfile(STAMP):
        return "", ""
    with open(STAMP) as fh:
        lines = fh.read().split()
    return (lines + ["", ""])[0], (lines + ["", ""])[1]


def is_stale() -> bool:
    """True unless the library on disk is the one the last build() wrote from the present sources and flags: the stamp
    holds the fingerprint of the sources AND the digest of the .so, so a library overwritten by some other tool (a
    partial link, a truncated copy, a foreign file) is stale even though the sources did not change."""
    if not os.path.isfile(LIB) or not os.path.isfile(STAMP):
        return True
    fp, digest = read_stamp()
    return fp != fingerprint() or digest != lib_digest()


def build_variant(out: str, extra: list, verbose: bool = False) -> str:
    """An experiment build with other flags into another file (tools/mkvariant.sh; selected with ICA_LIB=...)."""
    with tempfile.TemporaryDirectory(prefix="ica_build_") as objdir:
        cmds = commands(objdir, out, extra)
        with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as pool:
            list(pool.map(lambda c: subprocess.run(c, check=True, cwd=HERE, capture_output=not verbose), cmds[:-1]))
        subprocess.run(cmds[-1], check=True, cwd=HERE)
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not is_stale():
        return LIB
    fp = fingerprint()
    with tempfile.TemporaryDirectory(prefix="ica_build_") as objdir:
        cmds = commands(objdir)

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True, cwd=HERE)

        with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as pool:
            list(pool.map(run, cmds[:-1]))
        run(cmds[-1])
    with open(STAMP, "w") as fh:
        fh.write(fp + "\n" + lib_digest() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
