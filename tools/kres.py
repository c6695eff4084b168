#!/usr/bin/env python3
"""Compile libica_hip.so and print VGPR / occupancy / LDS per kernel (hipcc -Rpass-analysis=kernel-resource-usage)."""
import os, re, subprocess, sys
here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "inverse_compositional_algorithm_amd")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-o", os.path.join(here, "libica_hip.so"),
       os.path.join(here, "csrc", "ica_api.hip"), "-Rpass-analysis=kernel-resource-usage"]
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
