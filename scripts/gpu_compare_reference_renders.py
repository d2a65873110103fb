"""Compares this library's output with images the reference's authors rendered and checked in
(Renders/MultLightGlass-0.4-<spp>-<N|D>.ppm, kept as tests/golden/reference_renders/*.npz).
The file names give the scene, the resolution (0.4 x 1920x1080) and the sample count; gamma is
the reference's default 2.1 (PT.cpp:72); path depth and exposure were set in the GUI and are
not recorded, so the depth is scanned and the exposure fitted (one scalar).  For scale, the
same statistics between two independent renders of ours are printed beside each comparison."""
import glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import numpy as np
import ptk

G = 2.1
D = os.path.join(ROOT, "tests", "golden", "reference_renders")

def quantise(img, e):
    return np.floor(255.999 * (np.maximum(img, 0) / e) ** (1.0 / G)).astype(np.int32)

def stats(a, b):
    dd = np.abs(a - b)
    return (dd.mean(), 100 * (dd.max(axis=2) <= 4).mean(), 100 * (dd.max(axis=2) <= 8).mean(),
            100 * (dd.max(axis=2) <= 16).mean())

def block_diff(a, b, k=16):
    """Mean absolute difference of k x k block means, in levels: per-pixel noise averages out
    (by k), a bias does not."""
    H, W, _ = a.shape
    Hc, Wc = H // k * k, W // k * k
    ma = a[:Hc, :Wc].reshape(Hc // k, k, Wc // k, k, 3).mean(axis=(1, 3))
    mb = b[:Hc, :Wc].reshape(Hc // k, k, Wc // k, k, 3).mean(axis=(1, 3))
    return float(np.abs(ma - mb).mean())

def linear(ctx, H, W, spp):
    return ctx.read_planes(0, 3).reshape(3, H, W)[:, ::-1].transpose(1, 2, 0).astype(np.float64) / spp

def main():
    ctx = ptk.Context(0)
    scene = os.path.join(ROOT, "data", "scenes", "MultiLightGlass.scene")
    fits = {}
    for spp in (5000, 2500, 128):
        ref = np.load(os.path.join(D, "MultLightGlass-0.4-%d-N.npz" % spp))["rgb"].astype(np.int32)
        H, W, _ = ref.shape
        for depth in (4, 8, 10, 11, 12):
            raw, _, _ = ptk.scene_constants(scene, W, H, depth, 0.005, False)
            imgs = []
            for stream in (0, 1):
                ctx.set_constants(raw)
                ctx.set_stream_id(stream)
                ctx.render(spp, ptk.SEED_COUNTER)
                imgs.append(linear(ctx, H, W, spp))
            ok = (ref > 8).all(axis=2) & (imgs[0] > 1e-6).all(axis=2)
            e = float(np.exp(np.mean(np.log(imgs[0][ok]) - G * np.log((ref[ok] + 0.5) / 255.999))))
            q = [quantise(im, e) for im in imgs]
            s_ref, s_self = stats(q[0], ref), stats(q[0], q[1])
            fits[(spp, depth)] = e
            print("   block means (16x16): ours vs file %.3f levels, ours vs ours %.3f" % (block_diff(q[0], ref), block_diff(q[0], q[1])))
            print("N %4d spp depth=%d exposure %.4f: ours vs file %.3f levels, within 4/8/16: %.1f/%.1f/%.1f %%; ours vs ours %.3f levels, "
                  "%.1f/%.1f/%.1f %%; means file %s ours %s" % ((spp, depth, e) + s_ref + s_self + (
                      np.round(ref.mean(axis=(0, 1)), 2), np.round(q[0].mean(axis=(0, 1)), 2))), flush=True)

if __name__ == "__main__":
    main()
