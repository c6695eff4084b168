#!/usr/bin/env python3
"""Compile device code to ISA and print instruction statistics for one kernel (substring match on the mangled name)."""
import collections, os, re, subprocess, sys
here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "inverse_compositional_algorithm_amd")
pat = sys.argv[1] if len(sys.argv) > 1 else "accumulate2_kernelILi4ELi0ELi1E"
dump = len(sys.argv) > 2
os.makedirs("/tmp/asm", exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", "/tmp/asm/ica.s",
                os.path.join(here, "csrc", "ica_api.hip")], check=True, capture_output=True)
s = open("/tmp/asm/ica.s").read()
m = re.search(r"^(\S*%s\S*):" % re.escape(pat), s, re.M)
i = m.start(); j = s.index("s_endpgm", i)
body = [l.strip() for l in s[i:j].splitlines()]
body = [l for l in body if l and not l.startswith((";", ".")) and not l.endswith(":")]
c = collections.Counter(l.split()[0] for l in body)
g = collections.Counter()
for op, n in c.items():
    g["valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem"] += n
print(m.group(1)[:60], "total", len(body), dict(g))
print("barriers", c["s_barrier"], "waitcnt", c["s_waitcnt"], "branches", sum(n for o, n in c.items() if o.startswith("s_cbranch")))
print(c.most_common(28))
if dump:
    open("/tmp/asm/kernel_body.s", "w").write("\n".join(body))
    print("body -> /tmp/asm/kernel_body.s")
