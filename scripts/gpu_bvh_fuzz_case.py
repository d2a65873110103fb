"""One seed of gpu_bvh_fuzz.py taken apart: which pixels differ between the culling hierarchy and
PTK_SCENE_VISIT_ALL, and which of the two agrees with the index-order scan of the CPU checker
(correctly rounded libm).  Usage: gpu_bvh_fuzz_case.py SEED [spp]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import ptk                      # noqa: E402
from oracle import oracle as O  # noqa: E402
import gpu_bvh_fuzz as F        # noqa: E402


def main():
    seed = int(sys.argv[1])
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    W, H = 64, 48
    sc, what = F.bent_scene(seed, W, H)
    print(seed, what)
    O.build()
    ctx = ptk.Context(0)
    out = {}
    for name, flags in (("hierarchy", 0), ("visit_all", ptk.SCENE_VISIT_ALL)):
        st = ctx.set_scene(sc, want_features=1, flags=flags)
        for s in range(spp):
            rays, glo = ctx.render(1, ptk.SEED_REFERENCE, F.SEED if s == 0 else glo)
            out[name, s] = (ctx.read_planes().copy(), rays)
        print(name, st)
    with O.variant("R"):
        acc = O.new_accumulators(W, H)
        glo = F.SEED
        for s in range(spp):
            _, glo, rays = O.render_scene(sc, acc, 1, s, glo, want_features=True, threads=8)
            out["cpu", s] = (np.array(acc, copy=True), rays)
    same = lambda a, b: (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    for s in range(spp):
        a, ra = out["hierarchy", s]
        b, rb = out["visit_all", s]
        c, rc = out["cpu", s]
        c = c[:a.shape[0]] if c.shape[0] >= a.shape[0] else c
        print("after sample %d: rays hierarchy %d visit_all %d cpu %d" % (s + 1, ra, rb, rc))
        bad = np.nonzero(~same(a, b).all(axis=0))[0]
        for p in bad[:6]:
            n = min(a.shape[0], c.shape[0])
            print("  pixel %d (x %d y %d): hierarchy==cpu %s, visit_all==cpu %s" % (
                p, p % W, p // W, bool(same(a[:n, p], c[:n, p]).all()), bool(same(b[:n, p], c[:n, p]).all())))
            print("    hierarchy colour %s normal %s pos %s" % (a[0:3, p], a[3:6, p], a[12:15, p]))
            print("    visit_all colour %s normal %s pos %s" % (b[0:3, p], b[3:6, p], b[12:15, p]))
            print("    cpu       colour %s normal %s pos %s" % (c[0:3, p], c[3:6, p], c[12:15, p]))
    return 0


if __name__ == "__main__":
    sys.exit(main())
