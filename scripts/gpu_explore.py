"""Exploratory GPU-vs-oracle comparison; prints statistics (not a test)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import numpy as np
import ptk
from oracle import oracle as O

SEED = (0x9E3779B97F4A7C15, 0xD1B54A32D192ED03)

def cmp(name, a, b):
    a = a.astype(np.float64); b = b.astype(np.float64)
    both_nan = np.isnan(a) & np.isnan(b)
    exact = ((a == b) | both_nan)
    denom = np.maximum(np.abs(b), 1e-30)
    rel = np.where(exact, 0.0, np.abs(a - b) / denom)
    rel = np.where(np.isnan(rel), np.inf, rel)
    print("  %-10s exact %.5f  rel<1e-6 %.5f  rel<1e-4 %.5f  max rel %.3g  nan(a,b)=%d,%d" % (
        name, exact.mean(), (rel < 1e-6).mean(), (rel < 1e-4).mean(), rel.max(), np.isnan(a).sum(), np.isnan(b).sum()))

def main():
    ctx = ptk.Context(0)
    for scene in sys.argv[1:] or ["cornell", "cornell_crystal", "MultiLightGlass", "Box_Grid", "blur_test", "bars3"]:
        W, H = 128, 72
        sc = O.load_scene(os.path.join(ROOT, "data", "scenes", scene + ".scene"))
        consts = O.build_constants(sc, W, H, 4, 0.005, True)
        for spp in (1, 8):
            acc = O.new_accumulators(W, H)
            n, glo, rays = O.render(consts, acc, spp, 0, SEED, threads=8)
            for mode in (ptk.MATH_EXACT, ptk.MATH_FAST):
                ctx.set_constants(consts)
                ctx.set_math_mode(mode)
                t = time.time()
                grays, gglo = ctx.render(spp, ptk.SEED_REFERENCE, SEED)
                dt = time.time() - t
                g = ctx.read_planes()
                print("%s %dx%d spp=%d mode=%d rays %d vs %d glo_ok=%s kernel_ms=%.3f" % (
                    scene, W, H, spp, mode, grays, rays, gglo == glo, ctx.timing().render_ms))
                cmp("col", g[0:3], acc[0:3]); cmp("normal", g[3:6], acc[3:6]); cmp("alb1", g[6:9], acc[6:9])
                cmp("alb2", g[9:12], acc[9:12]); cmp("wpos", g[12:15], acc[12:15]); cmp("dl", g[15], acc[15])
                cmp("sdv", g[16:32], acc[16:32]); cmp("sd", g[32:38], acc[32:38])
                pix_exact = ((g[0:3] == acc[0:3]) | (np.isnan(g[0:3]) & np.isnan(acc[0:3]))).all(axis=0).mean()
                print("  pixels with bit-exact colour: %.5f" % pix_exact)
        # exposure + denoise on the oracle's accumulators
        ctx.set_math_mode(ptk.MATH_EXACT)
        ctx.set_constants(consts)
        ctx.write_planes(acc, n)
        for f in (0, 1, 2):
            print("  exposure f%d gpu %r oracle %r" % (f, ctx.auto_exposure(f), O.auto_exposure(acc, W, H, n, f)))
        w = O.load_weights(os.path.join(ROOT, "data", "weights", "BEST.txt"))
        feats, var = O.forward_prop(acc, W, H, n, w, threads=8)
        for N in (1, 3):
            den, wc = O.filter(acc, var, W, H, n, N, threads=8)
            for variant in (0, 1):
                r = ctx.denoise(w, N, variant, want_features=(variant == 0))
                print("  denoise N=%d variant=%d ms=%.3f" % (N, variant, ctx.timing().denoise_ms))
                if r["features"] is not None:
                    for k0, nm in ((0, "prim"), (5, "sd"), (10, "mean7"), (15, "sd7"), (20, "grad"), (25, "mdev"), (30, "mad"), (35, "L")):
                        cmp("f:" + nm, r["features"][:, k0:k0 + (5 if k0 < 35 else 1)], feats[:, k0:k0 + (5 if k0 < 35 else 1)])
                cmp("var", r["variances"], var); cmp("denoised", r["denoised"], den); cmp("wcSum", r["wcSum"], wc)

if __name__ == "__main__":
    main()
