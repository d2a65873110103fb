"""Path-trace rate against launch length and frame size (cornell_crystal, no features):
tells a per-launch overhead (rate rises with the samples per launch) from a sustained-load
effect (rate falls in long launches).  Usage: gpu_launch_length.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import ptk


def main():
    scene = os.path.join(ROOT, "data", "scenes", "cornell_crystal.scene")
    ctx = ptk.Context(0)
    for (W, H), spps in (((1920, 1080), (50, 200, 625, 2000)), ((3840, 2160), (13, 50, 156, 625)),
                         ((3840, 1080), (100, 312)), ((1920, 2160), (100, 312))):
        raw, _, _ = ptk.scene_constants(scene, W, H, 4, 0.005, False)
        ctx.set_constants(raw)
        ctx.render(4, ptk.SEED_COUNTER)
        for spp in spps:
            ts = []
            for _ in range(3):
                ctx.render(spp, ptk.SEED_COUNTER)
                ts.append(ctx.timing().render_ms)
            ts.sort()
            print("%dx%d spp %4d: median %9.3f ms  %.1f Msamples/s" % (W, H, spp, ts[1], W * H * spp / ts[1] / 1e3),
                  flush=True)


if __name__ == "__main__":
    main()
