"""Golden vectors at the BASELINE.json configuration sizes (build container only; needs /root/reference).

    python tests/golden/make_golden_configs.py

Same recipe and record layout as make_golden.py (reference drivers through oracle/ref_harness.py on
synthetic.make_pair inputs), at the image sizes, transform, error function and pyramid depth of BASELINE.json's
configs[0..4]: the first pairs of each configuration, plus configs[4] at quarter size (SURVEY 8d parity protocol).
Grayscale configurations are fed to the reference replicated to 3 channels (it accepts nothing else).
Takes ~50 minutes of CPU from scratch; the output is a few hundred kB of numbers.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import run_case  # noqa: E402
from oracle import ref_harness  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_configs.json")
# (name, transform, robust, nscales, H, W, C, seed, lambda, delta, clutter)
CASES = (
    [("configs0_pair%d" % i, 4, 0, 1, 256, 256, 1, 1000 + i, 0.0, 10, False) for i in range(2)]
    + [("configs1_pair%d" % i, 4, 0, 3, 512, 512, 1, 2000 + i, 0.0, 10, False) for i in range(4)]
    + [("configs2_pair%d" % i, 5, 0, 4, 1024, 1024, 1, 3000 + i, 0.0, 10, False) for i in range(4)]
    + [("configs3_pair%d" % i, 5, 3, 4, 1024, 1024, 1, 4000 + i, 0.0, 10, True) for i in range(4)]
    + [("configs4_quarter_pair%d" % i, 5, 3, 3, 270, 480, 3, 5100 + i, 0.0, 10, True) for i in range(8)]
    + [("configs4_pair%d" % i, 5, 3, 5, 1080, 1920, 3, 5000 + i, 0.0, 10, True) for i in range(2)]
)   # the first 4 pairs per configuration; configs[4]: 2 pairs at full size + 8 at 480 x 270


def main():
    ref = ref_harness.load()
    # runs already in the file are kept (the file grows by the missing cases only; delete it to regenerate everything)
    runs = json.load(open(OUT))["runs"] if os.path.isfile(OUT) else []
    have = {r["name"] for r in runs}
    for c in CASES:
        if c[0] in have:
            continue
        t0 = time.time()
        r = run_case(ref, c)
        runs.append(r)
        print(r["name"], [(s["scale"], len(s["iterations"])) for s in r["scales"]], "err", r["error"],
              "%.0fs" % (time.time() - t0), flush=True)
        order = {c[0]: i for i, c in enumerate(CASES)}
        runs.sort(key=lambda r: order.get(r["name"], 1 << 30))
        tmp = OUT + ".tmp"
        json.dump({"generator": "tests/golden/make_golden_configs.py",
                   "source": "reference drivers via oracle/ref_harness.py", "runs": runs}, open(tmp, "w"), indent=1)
        os.replace(tmp, OUT)


if __name__ == "__main__":
    main()
