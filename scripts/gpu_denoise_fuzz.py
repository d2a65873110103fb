"""Denoiser (features, MLP, cross-bilateral filter) against the CPU restatement on random
accumulators: image sizes from 1x1 up (smaller than the 7x7 and filter windows, around the tile,
strip and wave widths), every filter half-width 0..8, every weight file, sample counts that are
and are not powers of two, and value classes the rendered frames do not reach (exact zeros,
constant planes, large magnitudes, tiny deviations).  Discrete feature planes must be equal bit
for bit; smooth ones, bandwidths, filtered colour and weight sums within the tolerances of
tests/test_gpu_parity.py.  With variant 2 (the tolerance-held feature pass) the 7x7 mean and the
Sobel magnitude are compared within a tolerance as well; primaries, per-pixel deviations, the
"median getter" and 1/spp stay bit for bit.  The kind "nonfinite" (on request) sprinkles NaN and
infinities into a frame: variant 0 must then equal the restatement as everywhere else (NaN where
it has NaN); variant 2 may hold something else where the restatement's four smooth statistics
are not finite, and must agree in everything downstream.  Usage: gpu_denoise_fuzz.py [cases] [seed] [variant] [kinds, e.g. nonfinite]"""
import glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import ptk
from oracle import oracle as O


def same(a, b):
    return (a == b) | (np.isnan(a) & np.isnan(b))


def rel_ok(a, b, rel, abs_=1e-7):
    a = a.astype(np.float64); b = b.astype(np.float64)
    return same(a, b) | (np.abs(a - b) <= rel * np.abs(b) + abs_)


def random_planes(rng, W, H, spp, kind):
    P = W * H
    pl = np.zeros((38, P), np.float32)
    if kind == "noise":
        pl[0:16] = rng.uniform(-1, 1, (16, P)) * spp
    elif kind == "frame":            # the value ranges of a rendered frame
        pl[0:3] = rng.uniform(0, 1, (3, P)) * spp
        nrm = rng.normal(size=(3, P)); nrm /= np.linalg.norm(nrm, axis=0)
        pl[3:6] = nrm * spp
        pl[6:12] = rng.uniform(0, 1, (6, P)) * spp
        pl[12:15] = rng.uniform(-10, 10, (3, P)) * spp
        pl[15] = rng.choice([0, .25, .5, .75, 1.0], P) * spp
    elif kind == "flat":             # constant planes with a few outliers: flat windows, ties
        pl[0:16] = rng.uniform(-2, 2, (16, 1)) * spp
        hot = rng.random(P) < 0.05
        pl[0:16, hot] = rng.uniform(-2, 2, (16, int(hot.sum()))) * spp
    elif kind == "zeros":            # mostly exact zeros, signed
        pl[0:16] = rng.uniform(-1, 1, (16, P)) * spp
        pl[0:16][rng.random((16, P)) < 0.7] = 0.0
        pl[0:16][rng.random((16, P)) < 0.1] = -0.0
    elif kind == "large":
        pl[0:16] = rng.uniform(-1, 1, (16, P)) * spp * rng.choice([1e-6, 1.0, 1e4], (16, 1))
    elif kind == "nonfinite":        # a rendered frame's ranges with NaN and infinities sprinkled in
        pl = random_planes(rng, W, H, spp, "frame")
        for value in (np.nan, np.inf, -np.inf):
            pl[0:16][rng.random((16, P)) < 0.004] = value
        return pl
    pl[32:38] = rng.uniform(1e-4, 0.5, (6, P))
    if kind in ("flat", "zeros"):
        pl[32:38][rng.random((6, P)) < 0.3] = 0.0       # zero deviations: the 1e-6 floors decide
    return pl.astype(np.float32)


KINDS = ("noise", "frame", "flat", "zeros", "large")


def fuzz(cases, seed, ctx=None, verbose=True, variant=0, kinds=KINDS):
    """Runs `cases` cases; returns the descriptions of those with differences."""
    exact = ((0, 15), (20, 25), (30, 36)) if variant != 2 else ((0, 10), (30, 36))
    smooth = [("sd7", 15, 20, 2e-5), ("meandev", 25, 30, 2e-5)]
    if variant == 2:
        smooth += [("mean7", 10, 15, 2e-6), ("sobel", 20, 25, 2e-5)]
    rng = np.random.default_rng(seed)
    weight_files = sorted(glob.glob(os.path.join(ROOT, "data", "weights", "*.txt")))
    fixed = [(1, 1), (1, 9), (9, 1), (2, 2), (3, 5), (6, 6), (7, 7), (8, 8), (31, 9), (32, 8), (33, 9),
             (61, 4), (62, 5), (63, 10), (64, 8), (65, 17), (70, 14), (124, 3), (125, 16)]
    sc = O.load_scene(os.path.join(ROOT, "data", "scenes", "cornell_crystal.scene"))
    ctx = ctx or ptk.Context(0)
    bad = []
    for i in range(cases):
        W, H = fixed[i] if i < len(fixed) else (int(rng.integers(1, 140)), int(rng.integers(1, 90)))
        spp = int(rng.choice([1, 2, 3, 7, 8, 64, 100, 128, 1000, 4097]))
        kind = kinds[i % len(kinds)]
        wf = weight_files[i % len(weight_files)]
        w = O.load_weights(wf)
        pl = random_planes(rng, W, H, spp, kind)
        ctx.set_constants(O.build_constants(sc, W, H, 4, 0.005, True))
        ctx.write_planes(pl, spp)
        feats, var = O.forward_prop(pl, W, H, spp, w, threads=8)
        # Absolute floor of the comparisons: a few units in the last place of the largest mean
        # value.  The 3x3 mean is formed as sum * (1/9) on the device and as a sum of quotients
        # by the reference, so on a flat window |value - mean| is rounding noise of that size in
        # both; filtered colours of mixed sign cancel to the same level.
        floor = 4e-7 * max(1.0, float(np.abs(pl[0:16]).max()) / spp)
        msgs = []
        for n in ([1] + [int(x) for x in rng.choice([0, 2, 3, 4, 5, 6, 7, 8], 2, replace=False)]):
            den, wc = O.filter(pl, var, W, H, spp, n, threads=8)
            r = ctx.denoise(w, n, variant, want_features=True)
            f = r["features"]
            for lo, hi in exact:
                if not same(f[:, lo:hi], feats[:, lo:hi]).all():
                    msgs.append("N=%d features %d..%d differ in %d words" % (
                        n, lo, hi, int((~same(f[:, lo:hi], feats[:, lo:hi])).sum())))
            checks = [(nm, f[:, lo:hi], feats[:, lo:hi], tol) for nm, lo, hi, tol in smooth]
            checks += [("variances", r["variances"], var, 2e-5), ("denoised", r["denoised"], den, 5e-5),
                       ("wcSum", r["wcSum"], wc, 5e-5)]
            floored = ("meandev", "denoised", "wcSum") + (("sd7", "mean7", "sobel") if variant == 2 else ())
            for name, got, want, tol in checks:
                # On a flat window the reference's 7x7 deviation is |value - mean7| with mean7 a
                # sequential sum of 49 quotients: rounding noise of a few ulp of the value, where
                # the shared-statistics form gives (nearly) zero.  Three floors cover it.
                f = 3.0 * floor if (name == "sd7" and variant == 2) else floor
                ok = rel_ok(got, want, tol, f if name in floored else 1e-7)
                if variant == 2 and kind == "nonfinite" and name in ("sd7", "mean7", "sobel", "meandev"):
                    # the cheaper forms of these statistics reach an infinity where the reference's
                    # order of operations meets inf - inf first, and leave out the Sobel taps of
                    # weight zero, which the reference multiplies (0 * NaN): only the pixel's own
                    # value has weight zero in both gradients, and that pixel is NaN downstream
                    ok |= ~np.isfinite(want)
                if not ok.all():
                    j = np.argwhere(~ok)[0]
                    msgs.append("N=%d %s: %d outside %g, first at %s: %r vs %r" % (
                        n, name, int((~ok).sum()), tol, tuple(j), got[tuple(j)], want[tuple(j)]))
        if msgs:
            bad.append("case %d %dx%d spp %d %s %s: %s" % (i, W, H, spp, kind, os.path.basename(wf),
                                                         "; ".join(msgs[:6])))
            if verbose:
                print(bad[-1], flush=True)
        elif verbose and i % 10 == 0:
            print("case %d ok (%dx%d spp %d %s)" % (i, W, H, spp, kind), flush=True)
    return bad


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    bad = fuzz(cases, int(sys.argv[2]) if len(sys.argv) > 2 else 11,
               variant=int(sys.argv[3]) if len(sys.argv) > 3 else 0,
               kinds=tuple(sys.argv[4].split(",")) if len(sys.argv) > 4 else KINDS)
    print("%d cases, %d with differences" % (cases, len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
