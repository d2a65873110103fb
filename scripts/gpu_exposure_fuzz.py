"""The luminance reduce behind auto exposure against the CPU restatement on random colour planes:
image sizes from 1x1 up (pixel counts that are and are not multiples of four, smaller than a
wave, larger than one round of workgroups), the three formulas, value classes a rendered frame
does not reach (negative sums, exact zeros, huge and subnormal values, NaN and infinities).  The
raw sum must agree within 1e-9 relative (the per-pixel terms are the same fp32 values; only the
order of the double additions differs) or be of the same class where the restatement's is not
finite, and the exposure within 2e-6 plus, for the two formulas the reference evaluates from a
sequential fp32 sum, that sum's own error bound (the library scales the double sum).  Usage: gpu_exposure_fuzz.py [cases] [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
sys.path.insert(0, ROOT)
import ptk                      # noqa: E402
from oracle import oracle as O  # noqa: E402


def fuzz(cases, seed, ctx=None, verbose=True):
    rng = np.random.default_rng(seed)
    sc = O.load_scene(os.path.join(ROOT, "data", "scenes", "cornell_crystal.scene"))
    ctx = ctx or ptk.Context(0)
    fixed = [(1, 1), (2, 1), (3, 1), (4, 1), (5, 1), (63, 1), (64, 1), (65, 1), (255, 1), (256, 1), (257, 1),
             (1023, 3), (1024, 1), (1025, 1), (2048, 300), (1921, 1081)]
    bad = []
    for i in range(cases):
        W, H = fixed[i] if i < len(fixed) else (int(rng.integers(1, 700)), int(rng.integers(1, 300)))
        P = W * H
        spp = int(rng.choice([1, 3, 8, 100, 4097]))
        kind = ("frame", "signed", "edges", "nonfinite", "cancel")[i % 5]
        pl = np.zeros((38, P), np.float32)
        pl[0:3] = rng.uniform(0, 1.5, (3, P)) * spp
        if kind == "signed":
            pl[0:3] = rng.uniform(-2, 2, (3, P)) * spp
        elif kind == "edges":
            choices = np.float32([0.0, -0.0, 1.0, 1e-30, 1e-42, 1e30, 3e38, 2.0 ** -24])
            pl[0:3] = rng.choice(choices, (3, P))
        elif kind == "nonfinite":
            value = (np.nan, np.inf, -np.inf)[(i // 5) % 3]
            pl[0:3][rng.random((3, P)) < min(0.5, 2.0 / P + 0.001)] = value
        elif kind == "cancel":          # large terms of both signs that cancel in the sum
            half = P // 2
            big = rng.uniform(1e6, 1e9, (3, half)).astype(np.float32)
            pl[0:3, :half] = big
            pl[0:3, half:2 * half] = -big
        ctx.set_constants(O.build_constants(sc, W, H, 4, 0.005, False))
        ctx.write_planes(pl, spp)
        msgs = []
        for formula in (0, 1, 2):
            e_ref, raw_ref = O.auto_exposure(pl, W, H, spp, formula)
            e, raw = ctx.auto_exposure(formula)
            scale = float(np.abs(pl[0:3].astype(np.float64)).sum()) + 1e-300   # cancelling sums: against the terms
            if np.isfinite(raw_ref):
                ok = abs(raw - raw_ref) <= 1e-9 * scale
            else:
                ok = (raw == raw_ref) or (np.isnan(raw) and np.isnan(raw_ref))
            if not ok:
                msgs.append("formula %d raw sum %r vs %r" % (formula, raw, raw_ref))
            if np.isfinite(e_ref):
                # The reference forms formulas 0 and 1 from a sequential fp32 sum of the terms,
                # whose own rounding error is bounded by P * 2^-24 * sum |term|; the library
                # scales its double sum.  K: what the formula multiplies the sum by.
                K = (9.6 / W * H * spp, 9.6 / P * spp, 9.6 / (P * spp))[formula]
                slack = (P * 2.0 ** -24 if formula != 2 else 1e-9) * scale * K
                ok = abs(e - e_ref) <= 2e-6 * abs(e_ref) + slack + 1e-37
            else:
                ok = (e == e_ref) or (np.isnan(e) and np.isnan(e_ref))
            if not ok:
                msgs.append("formula %d exposure %r vs %r" % (formula, e, e_ref))
        if msgs:
            bad.append("case %d %dx%d spp %d %s: %s" % (i, W, H, spp, kind, "; ".join(msgs[:4])))
            if verbose:
                print(bad[-1], flush=True)
        elif verbose and i % 10 == 0:
            print("case %d ok (%dx%d spp %d %s)" % (i, W, H, spp, kind), flush=True)
    return bad


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    bad = fuzz(cases, int(sys.argv[2]) if len(sys.argv) > 2 else 3)
    print("%d cases, %d with differences" % (cases, len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
