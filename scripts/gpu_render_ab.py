"""A/B timing of the path-trace kernel: the default library against each library under
path-tracer_amd/variants/, alternating, several repetitions (median and spread of Msamples/s)."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, statistics
ROOT = %r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import ptk
W, H, spp = 1920, 1080, 64
ctx = ptk.Context(0)
for scene, feats in (("cornell_crystal", False), ("cornell", False), ("bars3", False), ("Box_Grid", False), ("MultiLightGlass", False), ("MultiLightGlass", True), ("cornell_crystal", True), ("Box_Grid", True)):
    raw, _, _ = ptk.scene_constants(os.path.join(ROOT, "data", "scenes", scene + ".scene"), W, H, 4, 0.005, feats)
    ctx.set_constants(raw); ctx.render(4, ptk.SEED_COUNTER)
    r = []
    for rep in range(7):
        ctx.render(spp, ptk.SEED_COUNTER); r.append(W*H*spp/ctx.timing().render_ms/1e3)
    print("  %%-16s feats=%%d  median %%.0f  min %%.0f max %%.0f Msamples/s" %% (scene, feats, statistics.median(r), min(r), max(r)), flush=True)
''' % ROOT
libs = [""] + sorted(glob.glob(os.path.join(ROOT, "path-tracer_amd", "variants", "*.so")))
if len(sys.argv) > 1 and sys.argv[1] == "segment-scans":   # the default library with and without its kernels for scenes that refract nothing
    libs = ["", "PTK_SEGMENT_SCANS"]
for rnd in range(2):
    for lib in libs:
        print("==", os.path.basename(lib) or "libptk.so (default)", "round", rnd, flush=True)
        env = dict(os.environ)
        if lib == "PTK_SEGMENT_SCANS": env[lib] = "1"
        elif lib: env["PTK_LIB"] = lib
        subprocess.run([sys.executable, "-c", code], env=env, timeout=300)
