"""One-off validation of the hierarchy path at a size the test suite does not afford: a grid scene
of side^3 shapes (default 100 -> 1 000 003 shapes) at 1920x1080, one reference-seeded pass on the
GPU, then `picks` random pixels recomputed on the CPU by the index-order scan over every shape.
Usage: python scripts/gpu_bvh_validate_large.py [side] [picks]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import ptk      # noqa: E402
import scenes   # noqa: E402
from oracle import oracle as O   # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n_picks = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
W, H = 1920, 1080
P = W * H
seed = (0x9E3779B97F4A7C15, 0xD1B54A32D192ED03)
sc = scenes.grid_scene(side, W, H)
ctx = ptk.Context(0)
st = ctx.set_scene(sc, want_features=1)
print("shapes %d, nodes %d, depth %d, host build %.0f ms" % (st["num_shapes"], st["nodes"], st["depth"],
                                                            st["build_ms"]), flush=True)
rays, glo = ctx.render(1, ptk.SEED_REFERENCE, seed)
gpu = ctx.read_planes(0, 16)
seeds, glo_o = O.seed_stream(seed, P)
assert glo == glo_o
picks = np.sort(np.random.default_rng(11).choice(P, n_picks, replace=False))
t0 = time.time()
ref = O.render_scene_pixels(sc, picks // W, picks % W, seeds[picks], want_features=True,
                            threads=int(os.environ.get("OMP_NUM_THREADS", "8")))
print("CPU index-order scan of %d pixels over %d shapes: %.1f s" % (n_picks, st["num_shapes"],
                                                                    time.time() - t0), flush=True)
got = gpu[:, picks].T
same = ((got == ref[:, :16]) | (np.isnan(got) & np.isnan(ref[:, :16]))).all(axis=1)
first = (got[:, 3:15] == ref[:, 3:15]).all(axis=1)
print("bit-identical pixels: %.4f (first-hit planes %.4f); rays of the frame %d" % (
    same.mean(), first.mean(), rays))
assert same.mean() >= 0.99 and first.mean() >= 0.999
print("ok")
