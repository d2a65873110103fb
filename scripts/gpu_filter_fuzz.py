"""Row-window filter kernel against the LDS-tile kernel (ptk_denoise variants 0 and 1) on many
image sizes around the strip width (62), the wave width (64) and tiny heights: every output word
must be identical, whole frame and one ragged band per size."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import numpy as np
import ptk

def main():
    rng = np.random.default_rng(7)
    sizes = [(w, h) for w in (1, 2, 31, 61, 62, 63, 64, 65, 123, 124, 125, 126, 186, 187, 247, 248, 249, 250, 311)
             for h in (1, 2, 3, 7, 8, 9, 16, 17, 33)]
    sizes += [(int(rng.integers(1, 700)), int(rng.integers(1, 300))) for _ in range(40)]
    w = ptk.load_weights(os.path.join(ROOT, "data", "weights", "BEST.txt"))
    ctx = ptk.Context(0)
    bad = 0
    for W, H in sizes:
        raw, _, _ = ptk.scene_constants(os.path.join(ROOT, "data", "scenes", "cornell_crystal.scene"), W, H, 4, 0.005, True)
        ctx.set_constants(raw)
        P = W * H
        spp = int(rng.integers(1, 300))
        pl = np.zeros((38, P), np.float32)
        pl[0:16] = (rng.uniform(-1, 1, (16, P)) * spp).astype(np.float32)
        pl[32:38] = rng.uniform(1e-4, 0.5, (6, P))
        ctx.write_planes(pl, spp)
        a = ctx.denoise(w, 1, 0)
        b = ctx.denoise(w, 1, 1)
        same = all(np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)) for k in ("denoised", "wcSum"))
        if not same:
            bad += 1
            print("DIFFERENT at %dx%d spp %d" % (W, H, spp), flush=True)
    print("%d sizes, %d different" % (len(sizes), bad))
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
