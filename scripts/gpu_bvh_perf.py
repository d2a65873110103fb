"""Throughput of the ptk_set_scene path (shapes in HBM behind a hierarchy) at several scene sizes,
and of the reference scenes through both paths.  Run on the GPU box: python scripts/gpu_bvh_perf.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import ptk      # noqa: E402
import scenes   # noqa: E402

W, H = 1920, 1080
c = ptk.Context(0)


def run(label, spp, feats=0):
    c.render(1, ptk.SEED_COUNTER)
    c.render(spp, ptk.SEED_COUNTER)
    t = c.timing()
    print("  %-34s feats=%d  %8.2f ms  %8.1f Msamples/s  %7.1f Mrays/s" % (
        label, feats, t.render_ms, t.render_samples / t.render_ms / 1e3,
        t.render_rays / t.render_ms / 1e3), flush=True)


extra = "--no-extra" not in sys.argv
sides = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [10, 22, 46, 100]
if sides == [0]:
    sides = []
print("library:", ptk.LIB_PATH, flush=True)
for side in sides:
    t0 = time.time()
    sc = scenes.grid_scene(side, W, H)
    t1 = time.time()
    st = c.set_scene(sc)
    print("grid side %d: %d shapes, %d nodes, depth %d, build %.0f ms (generate %.0f ms), %.1f MB" % (
        side, st["num_shapes"], st["nodes"], st["depth"], st["build_ms"], (t1 - t0) * 1e3,
        st["device_bytes"] / 1e6), flush=True)
    run("grid %d" % side, 8)
    c.set_scene(sc, want_features=1)
    run("grid %d" % side, 8, 1)

if not extra:
    sys.exit(0)
for n in (2000, 20000):
    for diel in (False, True):
        sc = scenes.random_scene(n, W, H, seed=n, dielectric=diel)
        st = c.set_scene(sc)
        print("random %d: depth %d, build %.0f ms" % (n, st["depth"], st["build_ms"]), flush=True)
        run("random %d%s" % (n, " (refracting shapes)" if diel else ""), 2)

raw, _, _ = ptk.scene_constants(os.path.join(ROOT, "data", "scenes", "cornell_crystal.scene"), W, H)
c.set_constants(raw, want_features=0)
run("cornell_crystal, fixed-size path", 32)
c.set_scene(ptk.scene_from_constants(raw))
run("cornell_crystal, through hierarchy", 32)
