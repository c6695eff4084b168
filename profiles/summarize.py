#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats kernel_stats.csv to the rows of this library."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
keep = [r for r in rows if "ica::" in r["Name"] or "rocclr" in r["Name"]]
tot = sum(float(r["TotalDurationNs"]) for r in keep)
print(f"{'kernel':<72} {'calls':>6} {'total_ms':>9} {'avg_us':>9} {'min_us':>8} {'max_us':>9} {'%ica':>6}")
for r in keep:
    name = r["Name"].replace("void ", "").split("(")[0][:72]
    print(f"{name:<72} {r['Calls']:>6} {float(r['TotalDurationNs'])/1e6:>9.2f} {float(r['AverageNs'])/1e3:>9.1f} "
          f"{float(r['MinNs'])/1e3:>8.1f} {float(r['MaxNs'])/1e3:>9.1f} {100*float(r['TotalDurationNs'])/tot:>6.1f}")
