"""Render throughput at maxDepth 4 / 8 / 12 (each a separate kernel instantiation)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import ptk
W, H, spp = 1920, 1080, 16
ctx = ptk.Context(0)
for scene in ("cornell_crystal", "MultiLightGlass"):
    for depth in (4, 8, 12):
        for feats in (False, True):
            raw, _, _ = ptk.scene_constants(os.path.join(ROOT, "data", "scenes", scene + ".scene"), W, H, depth, 0.005, feats)
            ctx.set_constants(raw); ctx.render(2, ptk.SEED_COUNTER)
            best = 1e30
            for rep in range(3):
                rays, _ = ctx.render(spp, ptk.SEED_COUNTER); best = min(best, ctx.timing().render_ms)
            print("%-16s depth %2d feats=%d  %.2f ms  %.0f Msamples/s  %.1f Grays/s" % (scene, depth, feats, best, W*H*spp/best/1e3, rays/best/1e6), flush=True)
