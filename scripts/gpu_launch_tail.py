"""How much of a path-trace launch is its tail?  A launch hands one 16x16-pixel workgroup per
tile to the hardware dispatcher, which keeps (workgroups per CU) x 256 of them resident; a
frame whose tile count is not a whole number of such rounds ends with a partly filled chip.
Renders cornell_crystal at 1920 x H for several H (same spp) and prints Msamples/s for each;
H = 1024 is exactly five rounds of 1536 workgroups, H = 1080 is 5.31.
Usage: gpu_launch_tail.py [spp] [H ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import ptk


def main():
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    heights = [int(x) for x in sys.argv[2:]] or [1024, 1080, 1232, 2048]
    scene = os.path.join(ROOT, "data", "scenes", "cornell_crystal.scene")
    ctx = ptk.Context(0)
    W = 1920
    for H in heights:
        raw, _, _ = ptk.scene_constants(scene, W, H, 4, 0.005, False)
        ctx.set_constants(raw)
        ctx.render(8, ptk.SEED_COUNTER)
        ts = []
        for _ in range(5):
            ctx.render(spp, ptk.SEED_COUNTER)
            ts.append(ctx.timing().render_ms)
        ts.sort()
        wgs = ((W + 15) // 16) * ((H + 15) // 16)
        print("1920x%d spp %d: %d workgroups = %.2f rounds of 1536; median %.3f ms, best %.3f ms, "
              "%.1f Msamples/s (median)" % (H, spp, wgs, wgs / 1536.0, ts[2], ts[0],
                                            W * H * spp / ts[2] / 1e3), flush=True)


if __name__ == "__main__":
    main()
