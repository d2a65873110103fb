"""Quick throughput probe (not the bench): render / exposure / denoise at 1080p."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import numpy as np
import ptk
from oracle import oracle as O

def main():
    W, H = 1920, 1080
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    scenes = sys.argv[2:] or ["cornell_crystal", "MultiLightGlass", "Box_Grid"]
    ctx = ptk.Context(0)
    for scene in scenes:
        sc = O.load_scene(os.path.join(ROOT, "data", "scenes", scene + ".scene"))
        for feats in (False, True):
            consts = O.build_constants(sc, W, H, 4, 0.005, feats)
            for mode in (ptk.MATH_EXACT, ptk.MATH_FAST):
                ctx.set_constants(consts)
                ctx.set_math_mode(mode)
                ctx.render(2, ptk.SEED_COUNTER)   # warm
                best = 1e30
                for rep in range(3):
                    rays, _ = ctx.render(spp, ptk.SEED_COUNTER)
                    best = min(best, ctx.timing().render_ms)
                print("%-16s feats=%d mode=%d spp=%d  %.2f ms  %.1f Msamples/s  %.1f Mrays/s" % (
                    scene, feats, mode, spp, best, W * H * spp / best / 1e3, rays / best / 1e3), flush=True)
        # denoise + exposure on last (features) accumulators, exact mode
        w = O.load_weights(os.path.join(ROOT, "data", "weights", "BEST.txt"))
        for N in (1, 3):
            for variant in (0, 1):
                ts = []
                for rep in range(5):
                    ctx.denoise(w, N, variant, want_outputs=False)
                    t = ctx.timing(); ts.append((t.denoise_ms, t.features_ms, t.filter_ms))
                b = min(ts)
                print("  denoise N=%d variant=%d: %.3f ms (features %.3f filter %.3f)" % (N, variant, b[0], b[1], b[2]), flush=True)
        es = []
        for rep in range(5):
            ctx.auto_exposure(0); es.append(ctx.timing().exposure_ms)
        print("  exposure: %.4f ms -> %.1f GB/s" % (min(es), W * H * 12 / min(es) / 1e6), flush=True)

if __name__ == "__main__":
    main()
