"""Post-process (display metrics 0..6, exposure, gamma, 8-bit quantisation) against the CPU
restatement on random accumulators: every image size from 1x1 up (pixel counts that are not
multiples of the four a lane handles), value classes a rendered frame does not reach (negative
sums, exact zeros, exactly one, tiny and huge values, NaN and infinities), random exposure and
gamma.  The float screen must agree within 2e-6 relative for results between 1e-3 and 1e3 and
2e-5 beyond (the power is exp2(y log2 v) on the hardware units, whose error grows with |log2 v|;
results below the smallest normal number may come out as zero), and must be of the same class where the restatement's is not finite; the
8-bit image must be the quantisation of the library's own float screen wherever that is finite
(the reference converts NaN and out-of-range values to int, which C++ leaves undefined).
Usage: gpu_post_process_fuzz.py [cases] [seed]"""
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
    w = O.load_weights(os.path.join(ROOT, "data", "weights", "BEST.txt"))
    ctx = ctx or ptk.Context(0)
    fixed = [(1, 1), (2, 1), (3, 1), (1, 3), (5, 1), (7, 3), (63, 1), (65, 3), (255, 1), (257, 3), (1023, 1)]
    bad = []
    for i in range(cases):
        W, H = fixed[i] if i < len(fixed) else (int(rng.integers(1, 300)), int(rng.integers(1, 40)))
        P = W * H
        spp = int(rng.choice([1, 3, 8, 100, 4097]))
        kind = ("frame", "signed", "edges", "nonfinite")[i % 4]
        pl = np.zeros((38, P), np.float32)
        pl[0:16] = rng.uniform(0, 1.5, (16, P)) * spp
        if kind == "signed":
            pl[0:16] = rng.uniform(-2, 2, (16, P)) * spp
        elif kind == "edges":
            choices = np.float32([0.0, -0.0, 1.0, 1e-30, 1e-42, 1e30, 255.0 / 255.999, 0.5, 2.0 ** -24])
            pl[0:16] = rng.choice(choices, (16, P)) * np.float32(spp)
        elif kind == "nonfinite":
            for value in (np.nan, np.inf, -np.inf):
                pl[0:16][rng.random((16, P)) < 0.02] = value
        pl[32:38] = rng.uniform(0, 0.5, (6, P))
        ctx.set_constants(O.build_constants(sc, W, H, 4, 0.005, True))
        ctx.write_planes(pl, spp)
        den = ctx.denoise(w, 1, 0)["denoised"]
        msgs = []
        for metric in range(7):
            exposure = float(rng.choice([1.0, 0.223, 2.0, rng.uniform(0.01, 8.0)]))
            g = float(rng.choice([2.2, 1.0, 2.1, rng.uniform(0.5, 3.0)]))
            ref = O.post_process(pl, den, W, H, spp, metric, exposure, g)
            post, rgb = ctx.post_process(metric, exposure, g, want_rgb8=True)
            fin = np.isfinite(ref)
            mid = (np.abs(ref) >= 1e-3) & (np.abs(ref) <= 1e3)
            with np.errstate(invalid="ignore", over="ignore"):
                ok = np.where(fin, np.abs(post.astype(np.float64) - ref) <= np.where(mid, 2e-6, 2e-5) * np.abs(ref) + 1.2e-38,
                              (post == ref) | (np.isnan(post) & np.isnan(ref)))
            if not ok.all():
                j = tuple(np.argwhere(~ok)[0])
                msgs.append("metric %d exposure %g g %g: %d values off, first %s: %r vs %r" % (
                    metric, exposure, g, int((~ok).sum()), j, post[j], ref[j]))
            own = O.quantise8(post, W, H)
            in_range = (np.isfinite(post) & (post >= 0) & (post < 256.0)).reshape(H, W, 3)[::-1]
            if not (rgb == own)[in_range].all():
                msgs.append("metric %d: 8-bit image is not the quantised float screen in %d places" % (
                    metric, int(((rgb != own) & in_range).sum())))
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
