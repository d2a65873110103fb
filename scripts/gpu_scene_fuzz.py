"""Random-scene parity fuzz: the check of tests/test_gpu_parity.py::test_random_scenes_... over
many seeds (default 400).  Prints every seed whose planes differ from the restatement.
Usage: gpu_scene_fuzz.py [scenes] [swap|degenerate]   ("swap": some boxes get lo > hi on one axis;
"degenerate": jitter off, box faces in the camera's planes, scenes far from the origin;
"sizes": a random frame size from 1x1 to 89x59 for every scene)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ptk
from oracle import oracle as O
import test_gpu_parity as T

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    swap = len(sys.argv) > 2 and sys.argv[2] == "swap"
    degenerate = len(sys.argv) > 2 and sys.argv[2] == "degenerate"
    sizes = len(sys.argv) > 2 and sys.argv[2] == "sizes"
    import numpy as np
    size_rng = np.random.default_rng(5)
    O.build()
    ctx = ptk.Context(0)
    bad = 0
    for seed in range(100, 100 + n):
        if sizes:       # frames from one pixel up, narrower and lower than a tile, not multiples of 16
            W, H = int(size_rng.integers(1, 90)), int(size_rng.integers(1, 60))
            r = T.random_scene_mismatch(ctx, O, seed, W=W, H=H, degenerate=bool(seed % 3 == 0))
        else:
            r = T.random_scene_mismatch(ctx, O, seed, swap_bounds=swap, degenerate=degenerate)
        if r:
            bad += 1
            print("seed %d: %s" % (seed, r[:300]), flush=True)
        if (seed - 99) % 50 == 0:
            print("%d scenes, %d differ" % (seed - 99, bad), flush=True)
    print("%d scenes, %d differ" % (n, bad))
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main())
