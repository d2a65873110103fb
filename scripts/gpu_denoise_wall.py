"""Host clock around ptk_denoise (argument checks, weight upload if changed, both launches, the wait)
next to the two kernels' event times, on a rendered MultiLightGlass frame.  PTK_LIB selects the library."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import ptk

def main():
    W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
    ctx = ptk.Context(0)
    raw, _, _ = ptk.scene_constants(os.path.join(ROOT, "data", "scenes", "MultiLightGlass.scene"), W, H, 4, 0.005, True)
    ctx.set_constants(raw)
    ctx.render(32, ptk.SEED_COUNTER)
    w = ptk.load_weights(os.path.join(ROOT, "data", "weights", "BEST.txt"))
    walls, kern = [], []
    for _ in range(60):
        t0 = time.perf_counter()
        ctx.denoise(w, 1, 0, want_outputs=False)
        walls.append((time.perf_counter() - t0) * 1e3)
        kern.append(ctx.timing().denoise_ms)
    walls, kern = sorted(walls[10:]), sorted(kern[10:])
    print("%dx%d ptk_denoise: host clock median %.4f ms (min %.4f), kernels median %.4f ms" % (
        W, H, walls[len(walls) // 2], walls[0], kern[len(kern) // 2]), flush=True)

if __name__ == "__main__":
    main()
