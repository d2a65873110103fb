"""Throughput of the hierarchy path on the benchmark lattice carried far from the origin
(every coordinate and the camera shifted by (4000, 250, -3000)): the bounds must not grow with
the distance from the origin.  Usage: gpu_bvh_far_scene.py [side]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import ptk      # noqa: E402
import scenes   # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 46
W, H = 1920, 1080
ctx = ptk.Context(0)
for shift in (np.float32([0, 0, 0]), np.float32([4000, 250, -3000])):
    sc = scenes.grid_scene(side, W, H)
    sa, st = sc["shape_attrs"], sc["shape_type"]
    sa[:, 0:3] += shift
    sa[st == 1, 3:6] += shift
    sa[st == 1, 6:9] += shift
    sc["cam_pos"] = (sc["cam_pos"] + shift).astype(np.float32)
    stats = ctx.set_scene(sc, want_features=0)
    ctx.render(1, ptk.SEED_COUNTER)
    ctx.render(8, ptk.SEED_COUNTER)
    t = ctx.timing()
    print("shift %s: %d shapes, depth %d, %.2f ms, %.1f Msamples/s" % (
        shift.tolist(), stats["num_shapes"], stats["depth"], t.render_ms, t.render_samples / t.render_ms / 1e3), flush=True)
