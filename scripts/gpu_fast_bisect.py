"""Runs Box_Grid in fast mode for each libptk variant in path-tracer_amd/variants (fast mode with
one of its parts switched back to exact arithmetic) and prints image mean and throughput."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
import numpy as np
ROOT = %r
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import ptk
W, H = 480, 270
ctx = ptk.Context(0)
raw, _, _ = ptk.scene_constants(os.path.join(ROOT, "data", "scenes", "Box_Grid.scene"), W, H, 4, 0.005, True)
for mode, name in ((ptk.MATH_EXACT, "exact"), (ptk.MATH_FAST, "fast")):
    ctx.set_constants(raw); ctx.set_math_mode(mode); ctx.render(1024, ptk.SEED_COUNTER)
    pl = ctx.read_planes(); ms = ctx.timing().render_ms
    print("  %%-5s mean %%.6f alb2 %%.5f dl %%.5f  %%.0f Msamples/s" %% (name, np.nanmean(pl[:3]) / 1024, np.nanmean(pl[9:12]) / 1024, np.nanmean(pl[15]) / 1024, W * H * 1024 / ms / 1e3), flush=True)
''' % ROOT
for lib in sorted(glob.glob(os.path.join(ROOT, "path-tracer_amd", "variants", "*.so"))) + [""]:
    print("==", os.path.basename(lib) or "libptk.so (default)", flush=True)
    env = dict(os.environ)
    if lib: env["PTK_LIB"] = lib
    subprocess.run([sys.executable, "-c", code], env=env, timeout=200)
