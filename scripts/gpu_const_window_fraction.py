import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import numpy as np
import ptk
W, H = 1920, 1080
ctx = ptk.Context(0)
for scene in ("MultiLightGlass", "cornell_crystal", "Box_Grid"):
    raw, _, _ = ptk.scene_constants(os.path.join(ROOT, "data", "scenes", scene + ".scene"), W, H, 4, 0.005, True)
    ctx.set_constants(raw); ctx.render(128, ptk.SEED_COUNTER)
    pl = ctx.read_planes(3, 13).reshape(13, H, W)
    groups = {"normal": pl[0:3], "alb1": pl[3:6], "alb2": pl[6:9], "wpos": pl[9:12], "dl": pl[12:13]}
    out = []
    flags = {}
    for name, g in groups.items():
        # wave footprint: rows 2r-1..2r+2 (4 rows), cols 32c-1..32c+32 (34 cols), clamped
        cnt = tot = 0
        flags[name] = []
        for r in range(0, H, 2):
            r0, r1 = max(r - 1, 0), min(r + 3, H)
            band = g[:, r0:r1, :]
            mn = band.min(axis=1); mx = band.max(axis=1)          # (ch, W)
            for c in range(0, W, 32):
                c0, c1 = max(c - 1, 0), min(c + 33, W)
                tot += 1
                flat = bool((mn[:, c0:c1].min(axis=1) == mx[:, c0:c1].max(axis=1)).all())
                flags[name].append(flat)
                if flat: cnt += 1
        out.append("%s %.1f%%" % (name, 100.0 * cnt / tot))
    both = np.array(flags["normal"]) & np.array(flags["alb1"])
    out.append("normal and alb1 together %.1f%%" % (100.0 * both.mean()))
    print(scene, " ".join(out), flush=True)
