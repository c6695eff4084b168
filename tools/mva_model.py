#!/usr/bin/env python3
"""The closed-network model of DESIGN.md 5.1: N workgroups per CU cycle between two processor-sharing stations, the tile
copies (demand D1) and the arithmetic + LDS reads (demand D2), both in microseconds per launch as measured by the
copies-only (ICA_ABLATE=1) and no-copies (ICA_ABLATE=2) builds.  Exact mean-value analysis; prints the launch time the
model predicts for N = 1 .. Nmax.

usage: mva_model.py D1_us D2_us [Nmax]        e.g.  mva_model.py 1835 2187   (one-channel robust homography, N = 6: 2376 us)
                                                    mva_model.py 3165 2708   (three channels, N = 4: 3693 us; measured 3667)"""
import sys


def mva(d1, d2, nmax):
    q1 = q2 = 0.0
    out = []
    for n in range(1, nmax + 1):
        r1, r2 = d1 * (1.0 + q1), d2 * (1.0 + q2)
        x = n / (r1 + r2)
        q1, q2 = x * r1, x * r2
        out.append((n, 1.0 / x))
    return out


if __name__ == "__main__":
    if len(sys.argv) < 3:
        sys.exit(__doc__)
    d1, d2 = float(sys.argv[1]), float(sys.argv[2])
    nmax = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    for n, t in mva(d1, d2, nmax):
        print(f"N = {n}: {t:8.1f} us per launch   (bound: {max(d1, d2):.1f})")
