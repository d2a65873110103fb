"""Runs scripts/gpu_perf.py-style render timing for each libptk variant in path-tracer_amd/variants."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
ROOT = %r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import ptk
W, H, spp = 1920, 1080, 128
ctx = ptk.Context(0)
for scene in ("cornell_crystal", "Box_Grid", "MultiLightGlass"):
  for feats in (False, True):
    raw, _, _ = ptk.scene_constants(os.path.join(ROOT, "data", "scenes", scene + ".scene"), W, H, 4, 0.005, feats)
    ctx.set_constants(raw); ctx.render(2, ptk.SEED_COUNTER)
    best = 1e30
    for rep in range(3):
        ctx.render(spp, ptk.SEED_COUNTER); best = min(best, ctx.timing().render_ms)
    print("  %%-16s feats=%%d  %%.2f ms  %%.0f Msamples/s" %% (scene, feats, best, W*H*spp/best/1e3), flush=True)
''' % ROOT
for lib in sorted(glob.glob(os.path.join(ROOT, "path-tracer_amd", "variants", "*.so"))) + [""]:
    print("==", os.path.basename(lib) or "libptk.so (default)", flush=True)
    env = dict(os.environ)
    if lib: env["PTK_LIB"] = lib
    subprocess.run([sys.executable, "-c", code], env=env, timeout=200)
