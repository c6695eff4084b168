#!/usr/bin/env python3
"""profiles/<tag>_* from what tools/refresh_profiles_gpu.sh <tag> left in gpurun_out/ (run here, after the GPU call)."""
import json, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
t = sys.argv[1]
go, pr = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")

def rd(name):
    return open(os.path.join(go, name)).read()

names = {"cfg2": ("cfg2_configs1", "--config 2"), "cfg3": ("cfg3_configs2", "--config 3"), "cfg4": ("cfg4_configs3_headline", "--config 4"),
         "cfg5": ("cfg5_configs4_shard", "--config 5"), "cfg4_u8": ("cfg4_configs3_uint8", "--config 4 --storage u8")}
for k, (nm, args) in names.items():
    body = rd(f"{t}_bs_{k}_summary.txt")
    line = rd(f"{t}_bs_{k}_line.json").strip().splitlines()[-1]
    with open(os.path.join(pr, f"{t}_bench_summary_{nm}.txt"), "w") as f:
        f.write(f"rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline --no-per-config {args} ... (tools/prof_bench.sh)\n\n")
        f.write(body)
        f.write("\nbench.py JSON line printed by the same (profiled) run:\n" + line + "\n")
cmds = {"h_rob": "--size 1024 1024 --transform 5 --robust 3 --batch 1024", "h_l2": "--size 1024 1024 --transform 5 --robust 0 --batch 1024",
        "a_l2": "--size 512 512 --transform 4 --robust 0 --batch 1024", "h_rob3": "--size 1080 1920 --channels 3 --transform 5 --robust 3 --batch 256"}
kernels = {}
for s, args in cmds.items():
    with open(os.path.join(pr, f"{t}_kbench_full_batch_{s}.txt"), "w") as f:
        f.write(f"rocprofv3 --kernel-trace --stats -- python3 tools/kbench.py --reps 5 {args}   (tools/prof_kbench.sh)\n\n")
        f.write(rd(f"{t}_kb_{s}.txt"))
    shutil.copy(os.path.join(go, f"{t}_pmc_{s}_fetch_size.csv"), os.path.join(pr, f"{t}_pmc_{s}_counters_raw.csv"))
    d = json.load(open(os.path.join(go, f"{t}_pmc_{s}_traffic.json")))
    for k, v in d["kernels"].items():
        key = "<" + k.split("<")[1].split(">")[0].rsplit(",", 1)[0] + ">"   # drop the storage flag: "<5, 1, 1, 3>"
        if k.rstrip(">").endswith("true"):
            key += " u8"
        v["source"] = f"profiles/{t}_pmc_{s}_counters_raw.csv"
        v["command"] = d["command"]
        kernels[key] = v
old = json.load(open(os.path.join(pr, "r02_traffic.json")))
json.dump({"what": old["what"] + "  library_fingerprint = the source fingerprint (build stamp) of the library the counters were "
           "collected on: bench.py reports roofline.traffic only when it is the library it has loaded.",
           "library_fingerprint": rd(f"{t}_fingerprint.txt").strip(), "kernels": kernels},
          open(os.path.join(pr, f"{t}_traffic.json"), "w"), indent=1)
line = rd(f"{t}_bench_line.json").strip().splitlines()[-1]
json.loads(line)
open(os.path.join(pr, f"{t}_bench_line.json"), "w").write(line + "\n")
print("assembled", sorted(x for x in os.listdir(pr) if x.startswith(t + "_")))
for k, v in kernels.items():
    print(k, "fetch/alg %.3f" % v["fetch_bytes_over_algorithmic_bytes"], "by size %.3f" % v.get("bytes_by_request_size_over_algorithmic", float("nan")))
