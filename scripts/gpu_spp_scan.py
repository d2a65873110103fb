"""Path-trace rate against the samples per launch, with and without features (cornell_crystal).
Usage: gpu_spp_scan.py [width height]"""
import os, sys
sys.path.insert(0, "path-tracer_amd")
import ptk
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
ctx = ptk.Context(0)
for feats in (False, True):
    raw, _, _ = ptk.scene_constants("data/scenes/cornell_crystal.scene", W, H, 4, 0.005, feats)
    ctx.set_constants(raw)
    ctx.render(4, ptk.SEED_COUNTER)
    for spp in (1, 2, 4, 8, 16, 32):
        ts = []
        for _ in range(7):
            ctx.render(spp, ptk.SEED_COUNTER); ts.append(ctx.timing().render_ms)
        ts.sort()
        print("features %d spp %2d: %.3f ms  %.0f Msamples/s" % (feats, spp, ts[3], W * H * spp / ts[3] / 1e3), flush=True)
