#!/bin/bash
# usage (GPU box): tools/pmc_traffic.sh <tag> [kbench args...]
# One rocprofv3 --pmc FETCH_SIZE pass (counters only, no tracing besides --kernel-trace) over tools/kbench.py: ONE
# full-batch launch shape of the fused iteration kernel.  Keeps the RAW counter CSV rows of the iteration kernels in
# gpurun_out/<tag>_fetch_size.csv (copied to profiles/ by hand) and prints bytes / algorithmic bytes.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmct_$tag
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmct_$tag -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --reps 3 "$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc.log 2>&1
python3 - "$tag" "$@" <<'PY'
import csv, glob, json, os, sys
tag = sys.argv[1]
args = sys.argv[2:]
root = os.environ["GRAFT_REPO_ROOT"]
def opt(name, n, default):
    return [type(default[0])(v) for v in args[args.index(name) + 1:args.index(name) + 1 + n]] if name in args else default
H, W = opt("--size", 2, [512, 512]); B = opt("--batch", 1, [1024])[0]; C = opt("--channels", 1, [1])[0]
alg = 2.0 * B * H * W * C * 4
rows = []
for fn in glob.glob(f"/tmp/pmct_{tag}/*/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        if "ica_iterate_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            rows.append(r)
out = os.path.join(root, "gpurun_out", f"{tag}_fetch_size.csv")
with open(out, "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
    w.writeheader(); w.writerows(rows)
by = {}
for r in rows:
    by.setdefault(r["Kernel_Name"].split("(")[0].replace("void ", ""), []).append(float(r["Counter_Value"]))
res = {}
for k, v in by.items():
    kb = sum(v) / len(v)
    # FETCH_SIZE is in KB; on gfx950 it counts 128-B requests as 64 B: x2 (MI355X_MICROARCH.md, section HBM)
    byts = kb * 1024 * 2
    # the Hessian-only kernel (mode 2) reads I1 only
    a = alg / 2 if ", 2, " in k else alg
    res[k] = {"FETCH_SIZE_KB_mean": kb, "launches": len(v), "bytes_x1024_x2": byts, "algorithmic_bytes": a,
              "fetch_bytes_over_algorithmic_bytes": byts / a}
    print(k, json.dumps(res[k]))
json.dump({"command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 tools/kbench.py --reps 3 " + " ".join(args),
           "kernels": res}, open(os.path.join(root, "gpurun_out", f"{tag}_traffic.json"), "w"), indent=1)
PY
