#!/bin/bash
# usage (GPU box): tools/pmc_traffic.sh <tag> [kbench args...]
# One rocprofv3 --pmc FETCH_SIZE pass (counters only, no tracing besides --kernel-trace) over tools/kbench.py: ONE
# full-batch launch shape of the fused iteration kernel.  Keeps the RAW counter CSV rows of the iteration kernels in
# gpurun_out/<tag>_fetch_size.csv (copied to profiles/ by hand) and prints bytes / algorithmic bytes.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmct_$tag
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmct_$tag -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --reps 3 "$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc.log 2>&1
# second pass (its own run): the L2's memory-side read requests by size -- an exact byte count, no correction factor
rm -rf /tmp/pmcr_$tag
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d /tmp/pmcr_$tag -- python3 $GRAFT_REPO_ROOT/tools/kbench.py --reps 3 "$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc2.log 2>&1
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
for fn in glob.glob(f"/tmp/pmct_{tag}/*/*counter_collection.csv") + glob.glob(f"/tmp/pmcr_{tag}/*/*counter_collection.csv"):
    for r in csv.DictReader(open(fn)):
        if "ica_iterate_kernel" in r["Kernel_Name"]:
            rows.append(r)
out = os.path.join(root, "gpurun_out", f"{tag}_fetch_size.csv")
with open(out, "w", newline="") as fh:
    w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
    w.writeheader(); w.writerows(rows)
by, sized = {}, {}
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if r["Counter_Name"] == "FETCH_SIZE":
        by.setdefault(k, []).append(float(r["Counter_Value"]))
    else:
        sized.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
res = {}
for k, v in by.items():
    kb = sum(v) / len(v)
    # FETCH_SIZE is in KB; on gfx950 it counts 128-B requests as 64 B: x2 (MI355X_MICROARCH.md, section HBM)
    byts = kb * 1024 * 2
    # the Hessian-only kernel (mode 2) reads I1 only
    a = alg / 2 if ", 2, " in k else alg
    res[k] = {"FETCH_SIZE_KB_mean": kb, "launches": len(v), "bytes_x1024_x2": byts, "algorithmic_bytes": a,
              "fetch_bytes_over_algorithmic_bytes": byts / a}
    if k in sized and all(n in sized[k] for n in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum")):
        m = {n: sum(x) / len(x) for n, x in sized[k].items()}
        exact = 32 * m["TCC_EA0_RDREQ_32B_sum"] + 64 * m["TCC_EA0_RDREQ_64B_sum"] + 128 * m["TCC_EA0_RDREQ_128B_sum"]
        res[k].update({"read_requests": m, "bytes_by_request_size": exact, "bytes_by_request_size_over_algorithmic": exact / a})
    print(k, json.dumps(res[k]))
json.dump({"command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 tools/kbench.py --reps 3 " + " ".join(args),
           "kernels": res}, open(os.path.join(root, "gpurun_out", f"{tag}_traffic.json"), "w"), indent=1)
PY
