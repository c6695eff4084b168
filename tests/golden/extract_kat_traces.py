"""Transcribe the per-iteration traces stored in the reference's notebooks into JSON.

Run in the build container only (needs /root/reference):
    python tests/golden/extract_kat_traces.py

The notebooks (test/inverse_compositional_algorithm.ipynb cells 14-16 and
test/inverse_compositional_algorithm_robust.ipynb cells 13-15) hold the stdout of the
reference drivers run with verbose=True by the reference's author on the pinned stack
(numpy 1.25 / scikit-image 0.24).  Only numbers are extracted: |Dp|, p and lambda per
iteration, grouped by run and by pyramid scale.  No source text is copied.
"""
import json
import os
import re

REF_TEST = "/root/reference/test"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "kat_traces.json")

IT_L2 = re.compile(r"^Iteration (\d+): \|Dp\|=(\S+): p=\(([^)]*)\)\s*$")
IT_RB = re.compile(r"^\|Dp\|=(\S+): p=\(([^)]*)\), lambda_=(\S+)\s*$")


def stream_text(cell):
    return "".join("".join(o["text"]) for o in cell.get("outputs", []) if o.get("output_type") == "stream")


def parse_runs(text):
    """Split a cell's stdout into runs (one per 'Processing dataset image') and scales."""
    runs, cur, scale = [], None, None
    for line in text.splitlines():
        m = re.match(r"^Processing dataset image:\s+(\S+)", line)
        if m:
            cur = {"sample": m.group(1), "scales": []}
            runs.append(cur)
            scale = None
            continue
        if cur is None:
            continue
        m = re.match(r"^Scale: (\d+)", line)
        if m:
            scale = {"scale": int(m.group(1)), "iterations": []}
            cur["scales"].append(scale)
            continue
        m = IT_L2.match(line)
        if m:
            if scale is None:
                scale = {"scale": 0, "iterations": []}
                cur["scales"].append(scale)
            scale["iterations"].append({"dp_norm": float(m.group(2)), "p": [float(v) for v in m.group(3).split()]})
            continue
        m = IT_RB.match(line)
        if m:
            if scale is None:
                scale = {"scale": 0, "iterations": []}
                cur["scales"].append(scale)
            scale["iterations"].append({"dp_norm": float(m.group(1)), "p": [float(v) for v in m.group(2).split()],
                                        "lambda": float(m.group(3))})
    return runs


def main():
    out = {}
    nb = json.load(open(os.path.join(REF_TEST, "inverse_compositional_algorithm.ipynb")))
    cells = nb["cells"]
    out["l2_single_translation"] = parse_runs(stream_text(cells[14]))   # cell 14
    out["l2_single_rotation"] = parse_runs(stream_text(cells[15]))      # cell 15
    out["l2_pyramid3"] = parse_runs(stream_text(cells[16]))             # cell 16
    nb = json.load(open(os.path.join(REF_TEST, "inverse_compositional_algorithm_robust.ipynb")))
    cells = nb["cells"]
    out["charbonnier_single_translation"] = parse_runs(stream_text(cells[13]))
    out["charbonnier_single_rotation"] = parse_runs(stream_text(cells[14]))
    out["charbonnier_pyramid3"] = parse_runs(stream_text(cells[15]))
    meta = {
        "source": "stored stdout of test/inverse_compositional_algorithm.ipynb (cells 14,15,16) and "
                  "test/inverse_compositional_algorithm_robust.ipynb (cells 13,14,15)",
        "settings": {"TOL": 1e-3, "nanifoutside": True, "delta": 10, "p0": 0, "nscales": 3, "nu": 0.5,
                     "robust": "CHARBONNIER", "lambda": 0.0},
    }
    json.dump({"meta": meta, "runs": out}, open(OUT, "w"), indent=1)
    for k, v in out.items():
        print(k, [(r["sample"], [(s["scale"], len(s["iterations"])) for s in r["scales"]]) for r in v])


if __name__ == "__main__":
    main()
