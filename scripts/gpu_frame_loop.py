"""The reference's interactive frame (PT.cpp:105-131): one sample per pixel, auto exposure,
denoise, post-process, every frame.  Wall time per call against the device time of its kernels
at 1920x1080: what the host side of each entry point costs.  Usage: gpu_frame_loop.py [frames]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import ptk  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 200
W, H = 1920, 1080
ctx = ptk.Context(0)
raw, exposure, g = ptk.scene_constants(os.path.join(ROOT, "data", "scenes", "cornell_crystal.scene"), W, H, 4, 0.005, True)
w = ptk.load_weights(os.path.join(ROOT, "data", "weights", "BEST.txt"))
ctx.set_constants(raw)
for _ in range(5):
    ctx.render(1, ptk.SEED_COUNTER); ctx.auto_exposure(0); ctx.denoise(w, 1, 0, want_outputs=False)
    ctx.post_process(1, 0.5, 2.2, want_post=False, want_rgb8=False)
wall = dict(render=0.0, exposure=0.0, denoise=0.0, post=0.0)
dev = dict(render=0.0, exposure=0.0, denoise=0.0, post=0.0)
t_all = time.perf_counter()
for _ in range(frames):
    t0 = time.perf_counter(); ctx.render(1, ptk.SEED_COUNTER); t1 = time.perf_counter()
    dev["render"] += ctx.timing().render_ms
    ctx.auto_exposure(0); t2 = time.perf_counter(); dev["exposure"] += ctx.timing().exposure_ms
    ctx.denoise(w, 1, 0, want_outputs=False); t3 = time.perf_counter(); dev["denoise"] += ctx.timing().denoise_ms
    ctx.post_process(1, 0.5, 2.2, want_post=False, want_rgb8=False); t4 = time.perf_counter()
    dev["post"] += ctx.timing().post_ms
    wall["render"] += t1 - t0; wall["exposure"] += t2 - t1; wall["denoise"] += t3 - t2; wall["post"] += t4 - t3
t_all = time.perf_counter() - t_all
for k in wall:
    print("%-9s wall %.3f ms per call, device %.3f ms" % (k, wall[k] / frames * 1e3, dev[k] / frames))
print("frame: %.3f ms wall (%.0f frames/s at 1 spp, features, denoise N=1, no read-back)" % (t_all / frames * 1e3, frames / t_all))
