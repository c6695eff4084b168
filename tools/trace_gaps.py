#!/usr/bin/env python3
"""Analyse a rocprofv3 kernel_trace.csv: busy time vs idle gaps per solve step of bench.py."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "ica::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# split into steps at ica_finish_kernel
steps, cur = [], []
for r in rows:
    cur.append(r)
    if "ica_finish_kernel" in r["Kernel_Name"]:
        steps.append(cur); cur = []
st = steps[-2] if len(steps) > 2 else steps[-1]
t0, t1 = int(st[0]["Start_Timestamp"]), int(st[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in st)
gaps = []
for a, b in zip(st[:-1], st[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    gaps.append((g, a["Kernel_Name"].split("(")[0][-30:], b["Kernel_Name"].split("(")[0][-30:]))
print(f"kernels {len(st)} span {(t1-t0)/1e3:.1f} us busy {busy/1e3:.1f} us idle {(t1-t0-busy)/1e3:.1f} us")
big = [g for g in gaps if g[0] > 8000]
print(f"gaps > 8us: {len(big)} total {sum(g[0] for g in big)/1e3:.1f} us; gaps <= 8us: {len(gaps)-len(big)} total {sum(g[0] for g in gaps if g[0] <= 8000)/1e3:.1f} us")
by = collections.Counter()
for g in big: by[(g[1], g[2])] += g[0]
for k, v in by.most_common(6): print(f"  {v/1e3:8.1f} us  after {k[0]} before {k[1]}")
per = collections.defaultdict(lambda: [0, 0])
for r in st:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")[-34:]
    per[n][0] += 1; per[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]): print(f"  {n:<36} x{c:<4} {t/1e3:8.1f} us")
if len(sys.argv) > 2:
    print("accumulate launches (us):", " ".join(f"{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:.0f}" for r in st if ("accumulate2" in r["Kernel_Name"] or "iterate" in r["Kernel_Name"])))
    print("grid sizes:", " ".join(r.get("Grid_Size_X", r.get("Grid_Size", "?")) for r in st if ("accumulate2" in r["Kernel_Name"] or "iterate" in r["Kernel_Name"])))
