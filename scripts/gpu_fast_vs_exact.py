"""PTK_MATH_FAST against PTK_MATH_EXACT: throughput, and image statistics against a
high-sample exact reference (relative MSE per pixel, mean luminance)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import ptk
W, H = 480, 270
ctx = ptk.Context(0)
for scene in ("cornell_crystal", "MultiLightGlass", "Box_Grid", "cornell"):
    raw, _, _ = ptk.scene_constants(os.path.join(ROOT, "data", "scenes", scene + ".scene"), W, H, 4, 0.005, True)
    ctx.set_constants(raw); ctx.set_math_mode(ptk.MATH_EXACT); ctx.set_stream_id(99)
    ctx.render(8192, ptk.SEED_COUNTER)
    ref = ctx.read_planes()[:16] / 8192.0
    for mode, name in ((ptk.MATH_EXACT, "exact"), (ptk.MATH_FAST, "fast")):
        ctx.set_constants(raw); ctx.set_math_mode(mode); ctx.set_stream_id(0)
        ctx.render(256, ptk.SEED_COUNTER)
        ms = ctx.timing().render_ms
        img = ctx.read_planes()[:16] / 256.0
        relmse = np.nanmean((img[:3] - ref[:3]) ** 2 / (ref[:3] ** 2 + 1e-3))
        feat = [float(np.nanmean(np.abs(img[a:b] - ref[a:b]))) for a, b in ((3, 6), (6, 9), (9, 12), (12, 15), (15, 16))]
        print("%-16s %-5s %.0f Msamples/s  mean %.6f (ref %.6f)  relMSE %.5f  feature L1 %s" % (
            scene, name, W * H * 256 / ms / 1e3, np.nanmean(img[:3]), np.nanmean(ref[:3]), relmse,
            " ".join("%.2e" % f for f in feat)), flush=True)
