"""Path-trace rate against the work items' chunk length (PTK_RENDER_CHUNK_SPP; 0 = the
launcher's choice): cornell_crystal 1080p, 200 spp per launch, and MultiLightGlass with features, 128 spp."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import os, sys
sys.path.insert(0, os.path.join(%r, "path-tracer_amd"))
import ptk
ctx = ptk.Context(0)
for scene, feats, spp in (("cornell_crystal", False, 200), ("MultiLightGlass", True, 128)):
    raw, _, _ = ptk.scene_constants(os.path.join(%r, "data", "scenes", scene + ".scene"), 1920, 1080, 4, 0.005, feats)
    ctx.set_constants(raw)
    ctx.render(8, ptk.SEED_COUNTER)
    ts = []
    for _ in range(5):
        ctx.render(spp, ptk.SEED_COUNTER)
        ts.append(ctx.timing().render_ms)
    ts.sort()
    print("  %%-16s feats=%%d spp %%d: median %%.3f ms  %%.1f Msamples/s" %% (scene, feats, spp, ts[2], 1920 * 1080 * spp / ts[2] / 1e3), flush=True)
''' % (ROOT, ROOT)
for chunk in sys.argv[1:] or ["0", "8", "16", "24", "32", "48", "64", "1000"]:
    env = dict(os.environ)
    env["PTK_RENDER_CHUNK_SPP"] = chunk
    if chunk == "0":
        env.pop("PTK_RENDER_CHUNK_SPP")
    print("chunk %s" % chunk, flush=True)
    subprocess.run([sys.executable, "-c", CODE], env=env, timeout=120)
