import os, sys
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import ptk
W, H = 480, 270
ctx = ptk.Context(0)
for scene in ("Box_Grid", "cornell_crystal"):
    raw, _, _ = ptk.scene_constants(os.path.join(ROOT, "data", "scenes", scene + ".scene"), W, H, 4, 0.005, True)
    for mode, name in ((ptk.MATH_EXACT, "exact"), (ptk.MATH_FAST, "fast")):
        ctx.set_constants(raw); ctx.set_math_mode(mode)
        ctx.render(64, ptk.SEED_COUNTER)
        pl = ctx.read_planes()
        print(scene, name, "NaN pixels colour", int(np.isnan(pl[0]).sum()), "normal", int(np.isnan(pl[3]).sum()),
              "mean", float(np.nanmean(pl[:3]))/64, "alb2 mean", float(np.nanmean(pl[9:12]))/64, "dl", float(np.nanmean(pl[15]))/64, "inf", int(np.isinf(pl[:3]).sum()))
