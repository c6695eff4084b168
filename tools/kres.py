#!/usr/bin/env python3
"""usage: kres.py [kernel name substring] [csrc file] [extra hipcc flags, e.g. -DICA_INST_TT=5]
Compile one translation unit of the library into a scratch directory and print VGPR / occupancy / LDS per kernel (hipcc -Rpass-analysis=kernel-resource-usage)."""
import os, re, subprocess, sys, tempfile
here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "inverse_compositional_algorithm_amd")
# (the object goes to a scratch directory: a tool must never write to the product library's path)
src = os.path.join(here, "csrc", sys.argv[2] if len(sys.argv) > 2 else "ica_api.hip")
with tempfile.TemporaryDirectory(prefix="ica_kres_") as tmp:
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", "-o", os.path.join(tmp, "kres.o"),
           src, "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:]
    out = subprocess.run(cmd, capture_output=True, text=True)
txt = out.stderr
if out.returncode:
    print("\n".join(l for l in txt.splitlines() if "error" in l or "note:" in l)[:4000]); sys.exit(1)
pat = sys.argv[1] if len(sys.argv) > 1 else "accumulate2"
cur = None
for line in txt.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        continue
    if cur is None:
        continue
    for key, rx in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r" SGPRs: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                    ("spill", r"VGPRs Spill: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        mm = re.search(rx, line)
        if mm:
            cur[key] = int(mm.group(1))
    if "lds" in cur:
        if pat in cur["name"]:
            t = re.search(r"ILi(\d)ELi(\d)ELi(\d)E", cur["name"])
            tag = f"TT={t.group(1)} MODE={t.group(2)} C={t.group(3)}" if t else cur["name"][:50]
            print(f"{tag:<28} vgpr={cur.get('vgpr')} sgpr={cur.get('sgpr')} occ={cur.get('occ')} spill={cur.get('spill')} scratch={cur.get('scratch')} lds={cur.get('lds')}")
        cur = None
