"""Denoiser micro-benchmark on synthetic accumulators (no render): seed 1234, colour and albedo in U[0,1],
normalised normal, worldPos in U[-10,10], directLight in {0, 1/4, .., 1}, stdDev in U[1e-4, 0.5], 128 spp."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import numpy as np
import ptk

def synthetic(W, H, spp=128, seed=1234):
    rng = np.random.default_rng(seed)
    P = W * H
    pl = np.zeros((38, P), np.float32)
    pl[0:3] = rng.uniform(0, 1, (3, P)) * spp
    nrm = rng.normal(size=(3, P)); nrm /= np.linalg.norm(nrm, axis=0)
    pl[3:6] = nrm * spp
    pl[6:9] = rng.uniform(0, 1, (3, P)) * spp
    pl[9:12] = rng.uniform(0, 1, (3, P)) * spp
    pl[12:15] = rng.uniform(-10, 10, (3, P)) * spp
    pl[15] = rng.choice([0, .25, .5, .75, 1.0], P) * spp
    pl[32:38] = rng.uniform(1e-4, 0.5, (6, P))
    return pl

def main():
    W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    real = len(sys.argv) > 4 and sys.argv[4] == "real"     # a rendered frame instead of noise
    ctx = ptk.Context(0)
    raw, _, _ = ptk.scene_constants(os.path.join(ROOT, "data", "scenes", "MultiLightGlass.scene"), W, H, 4, 0.005, True)
    ctx.set_constants(raw)
    if real:
        ctx.render(128, ptk.SEED_COUNTER)
    else:
        ctx.write_planes(synthetic(W, H), 128)
    w = ptk.load_weights(os.path.join(ROOT, "data", "weights", "BEST.txt"))
    for N in (1, 3):
        for variant in ((0, 2) if N == 1 else (0,)):
            ts = []
            for _ in range(reps):
                ctx.denoise(w, N, variant, want_outputs=False)
                t = ctx.timing(); ts.append((t.denoise_ms, t.features_ms, t.filter_ms))
            ts = ts[2:]
            a = [sum(x[i] for x in ts) / len(ts) for i in range(3)]
            print("%dx%d N=%d variant=%d: %.4f ms (features %.4f -> %.0f GB/s, filter %.4f -> %.0f GB/s; fused-equiv %.0f GB/s)" % (
                W, H, N, variant, a[0], a[1], W*H*100/max(a[1],1e-9)/1e6, a[2], W*H*132/max(a[2],1e-9)/1e6, W*H*104/a[0]/1e6), flush=True)
    es = []
    for _ in range(reps):
        ctx.auto_exposure(0); es.append(ctx.timing().exposure_ms)
    e = sum(es[2:]) / len(es[2:])
    print("%dx%d exposure: %.4f ms -> %.0f GB/s" % (W, H, e, W*H*12/e/1e6), flush=True)

if __name__ == "__main__":
    main()
