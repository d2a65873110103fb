"""Is Renders/EdgeBlur.ppm (384x216, the cornell scene by the look of it) a render this library
reproduces?  The file name records nothing: sample count, path depth and exposure are scanned
or fitted.  For each (depth, spp) two independently seeded renders of Scenes/cornell.scene are
compared with the file and with each other:

  exposure  one fitted scalar
  pix/self  mean |difference| per pixel and channel in 8-bit levels: ours vs file | ours vs ours
  blk/self  mean |difference| of 16x16 block means: bias | its noise floor
  hf        RMS of (image - its 5x5 box mean): ours | file (matches when the sample counts do)

Usage: gpu_compare_reference_edgeblur.py > profiles/r03_reference_edgeblur_scan.txt"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import numpy as np
from scipy.ndimage import uniform_filter
import ptk

G = 2.1
D = os.path.join(ROOT, "tests", "golden", "reference_renders")


def quantise(img, e):
    return np.floor(255.999 * (np.maximum(img, 0) / e) ** (1.0 / G)).astype(np.int32)


def block_diff(a, b, k=16):
    H, W, _ = a.shape
    Hc, Wc = H // k * k, W // k * k
    ma = a[:Hc, :Wc].reshape(Hc // k, k, Wc // k, k, 3).mean(axis=(1, 3))
    mb = b[:Hc, :Wc].reshape(Hc // k, k, Wc // k, k, 3).mean(axis=(1, 3))
    return float(np.abs(ma - mb).mean())


def hf(img):
    img = img.astype(np.float64)
    return float(np.sqrt(np.mean((img - uniform_filter(img, size=(5, 5, 1), mode="nearest")) ** 2)))


def main():
    ref = np.load(os.path.join(D, "EdgeBlur.npz"))["rgb"].astype(np.int32)
    H, W, _ = ref.shape
    scene = os.path.join(ROOT, "data", "scenes", sys.argv[1] if len(sys.argv) > 1 else "cornell.scene")
    ctx = ptk.Context(0)
    print("file: %dx%d, channel means %s, hf %.3f" % (W, H, np.round(ref.mean(axis=(0, 1)), 2), hf(ref)))
    print("%5s %5s %9s %8s %8s %8s %8s %8s  %s" % ("depth", "spp", "exposure", "pix", "self", "blk", "self", "hf ours", "channel means ours"))
    fine = len(sys.argv) > 2 and sys.argv[2] == "fine"
    for depth in ((8, 10, 11, 12) if fine else (2, 3, 4, 5, 6, 8, 12)):
        raw, _, _ = ptk.scene_constants(scene, W, H, depth, 0.005, False)
        for spp in ((24, 28, 30, 32, 34, 36, 40, 48, 64) if fine else (1, 2, 3, 4, 6, 8, 12, 16, 32)):
            imgs = []
            for stream in (0, 1):
                ctx.set_constants(raw)
                ctx.set_stream_id(stream)
                ctx.render(spp, ptk.SEED_COUNTER)
                imgs.append(ctx.read_planes(0, 3).reshape(3, H, W)[:, ::-1].transpose(1, 2, 0).astype(np.float64) / spp)
            # Exposure: the one scalar that brings the block means of our quantised image closest
            # to the file's.  (Both sides are then means of gamma-mapped noisy pixels, so the
            # downward pull that noise exerts on such a mean is the same when the sample counts
            # match; fitting in the linear domain would not see it.)
            lin_ref = ((ref + 0.5) / 255.999) ** G
            e0 = float(imgs[0].mean() / lin_ref.mean())
            cands = [e0 * (0.90 + 0.005 * i) for i in range(41)]
            e = min(cands, key=lambda c: block_diff(quantise(imgs[0], c), ref))
            q = [quantise(im, e) for im in imgs]
            print("%5d %5d %9.4f %8.3f %8.3f %8.3f %8.3f %8.3f  %s" % (
                depth, spp, e, np.abs(q[0] - ref).mean(), np.abs(q[0] - q[1]).mean(),
                block_diff(q[0], ref), block_diff(q[0], q[1]), hf(q[0]),
                np.round(q[0].mean(axis=(0, 1)), 2)), flush=True)


if __name__ == "__main__":
    main()
