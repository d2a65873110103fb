"""Soak: contexts created and destroyed, frame sizes, scenes and entry points changed at random for
many rounds; every call must succeed, a fixed probe render must give the same bits at the end as
at the start, and device memory in use with every context closed must be the same after all the
rounds as after half of them (nothing the library allocates outlives its context; the first
half is the warm-up in which the HIP runtime loads the kernels and sizes its scratch memory for
them, which it keeps for the life of the process: about 0.6 GB here).
Usage: gpu_soak.py [rounds] [seed]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import ptk       # noqa: E402
import scenes    # noqa: E402

SCENES = ["cornell", "cornell_crystal", "MultiLightGlass", "Box_Grid", "blur_test", "bars3"]
SEED = (0x9E3779B97F4A7C15, 0xD1B54A32D192ED03)


def used_mb():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info(0)
    return (total - free) / 1e6


def probe(ctx):
    raw, _, _ = ptk.scene_constants(os.path.join(ROOT, "data", "scenes", "cornell_crystal.scene"), 96, 54)
    ctx.set_constants(raw, want_features=1)
    ctx.render(3, ptk.SEED_REFERENCE, SEED)
    return ctx.read_planes().copy()


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    torch.zeros(1, device="cuda:0")
    weights = ptk.load_weights(os.path.join(ROOT, "data", "weights", "BEST.txt"))
    c0 = ptk.Context(0)
    first = probe(c0)
    # the kernels with the most scratch memory per lane, once, so that the runtime has sized its
    # scratch for them before the first reading below
    for depth in (4, 8, 12):
        c0.set_scene(scenes.random_scene(60, 32, 32, seed=3, dielectric=True, blur=True, max_depth=depth),
                     want_features=1)
        c0.render(1, ptk.SEED_COUNTER)
    c0.close()
    start = used_mb()
    peak = start
    ctx = None
    mid = None
    for r in range(rounds):
        if r == rounds // 2:
            if ctx is not None:
                ctx.close()
                ctx = None
            mid = used_mb()
        if ctx is None or rng.random() < 0.3:
            if ctx is not None:
                ctx.close()
            ctx = ptk.Context(0)
        W, H = int(rng.integers(1, 400)), int(rng.integers(1, 300))
        feats = int(rng.random() < 0.7)
        if rng.random() < 0.5:
            name = SCENES[int(rng.integers(len(SCENES)))]
            raw, _, _ = ptk.scene_constants(os.path.join(ROOT, "data", "scenes", name + ".scene"), W, H,
                                            max_depth=int(rng.choice([1, 4, 8, 12])))
            ctx.set_constants(raw, want_features=feats)
        else:
            n = int(rng.choice([0, 1, 60, 500, 4000, 30000]))
            sc = scenes.random_scene(n, W, H, seed=int(rng.integers(1 << 30)), dielectric=bool(rng.random() < 0.5),
                                     blur=bool(rng.random() < 0.3), max_depth=int(rng.choice([2, 4, 8])))
            ctx.set_scene(sc, want_features=feats)
        if rng.random() < 0.5:
            ctx.render(int(rng.integers(1, 6)), ptk.SEED_COUNTER)
        else:
            ctx.render(int(rng.integers(1, 6)), ptk.SEED_REFERENCE, SEED)
        if rng.random() < 0.5:
            ctx.auto_exposure(int(rng.integers(3)))
        if feats and rng.random() < 0.7:
            ctx.denoise(weights, n=int(rng.integers(0, 6)), variant=int(rng.choice([0, 1, 2])),
                        want_outputs=bool(rng.random() < 0.5), want_features=bool(rng.random() < 0.3))
            if W >= H and rng.random() < 0.3:
                ctx.save_target()
                ctx.train_step(weights, n=1)
        if rng.random() < 0.6:
            ctx.post_process(metric=int(rng.choice([0, 2, 3, 4, 5, 6])) if feats else 0, exposure=1.0, g=2.2,
                             want_rgb8=bool(rng.random() < 0.5))
        if rng.random() < 0.2:
            ctx.reset()
        peak = max(peak, used_mb())
        if (r + 1) % 25 == 0:
            print("%d rounds, %.0f MB in use (start %.0f, peak %.0f)" % (r + 1, used_mb(), start, peak), flush=True)
    ctx.close()
    end = used_mb()
    c1 = ptk.Context(0)
    last = probe(c1)
    c1.close()
    same = ((first.view(np.uint32) == last.view(np.uint32)) | (np.isnan(first) & np.isnan(last))).all()
    print("device memory in use with no context: %.1f MB at the start, %.1f MB after %d rounds, %.1f MB after %d; "
          "peak %.1f MB; probe render %s" % (start, mid, rounds // 2, end, rounds, peak,
                                           "same bits" if same else "DIFFERS"))
    return 0 if same and end <= mid + 8.0 else 1


if __name__ == "__main__":
    sys.exit(main())
