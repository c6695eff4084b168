#!/usr/bin/env python3
"""Per-loop instruction mix of one kernel: for every backward branch, count the instructions between its target label
and the branch (innermost loops show up as the short ones).  usage: isa_loops.py <kernel substring> [extra hipcc flags]"""
import collections, os, re, subprocess, sys
here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "inverse_compositional_algorithm_amd")
pat = sys.argv[1] if len(sys.argv) > 1 else "accumulate2_kernelILi4ELi0ELi1E"
os.makedirs("/tmp/asm", exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", "/tmp/asm/ica.s",
                os.path.join(here, "csrc", "ica_api.hip")] + sys.argv[2:], check=True, capture_output=True)
s = open("/tmp/asm/ica.s").read()
m = re.search(r"^(\S*%s\S*):" % re.escape(pat), s, re.M)
i = m.start(); j = s.index("s_endpgm", i)
lines = [l.strip() for l in s[i:j].splitlines()]
lines = [l for l in lines if l and not l.startswith((";", ".")) or re.match(r"^\.LBB\S+:", l)]
pos = {}
for k, l in enumerate(lines):
    mm = re.match(r"^(\.LBB\S+):", l)
    if mm: pos[mm.group(1)] = k
for k, l in enumerate(lines):
    mm = re.match(r"^s_cbranch\S*\s+(\.LBB\S+)|^s_branch\s+(\.LBB\S+)", l)
    if not mm: continue
    tgt = mm.group(1) or mm.group(2)
    if tgt in pos and pos[tgt] < k:
        body = [x for x in lines[pos[tgt]:k + 1] if not x.endswith(":")]
        c = collections.Counter(x.split()[0] for x in body)
        valu = sum(n for o, n in c.items() if o.startswith("v_"))
        pk = sum(n for o, n in c.items() if o.startswith("v_pk_"))
        lds = sum(n for o, n in c.items() if o.startswith("ds_"))
        vm = sum(n for o, n in c.items() if o.startswith(("buffer_", "global_")))
        print(f"loop {tgt:<12} instrs={len(body):4d} valu={valu:4d} (pk={pk}) salu={sum(n for o,n in c.items() if o.startswith('s_')):3d} lds={lds:3d} vmem={vm:3d}")
        if len(sys.argv) > 1 and os.environ.get("ISA_TOP"):
            print("   ", [(o, n) for o, n in c.most_common(40) if o.startswith("v_")])
