"""Statistics for comparing a render with an image the reference's authors rendered and
checked in (tests/golden/reference_renders/: Renders/MultLightGlass-0.4-<spp>-N.ppm).

What the file names and the reference's sources fix: the scene, 768x432 (resPerc 0.4 of
1920x1080, PT.cpp:53-54), the sample count, no denoising, gamma 2.1 (PT.cpp:72), randSamp 0.005
(PT.cpp:79), the post-process (c / exposure)^(1/g) and PT::SaveImage's int(255.999 c) without
clamping (PT.cpp:484-501).  What they do not record: the exposure and the path depth set in the
GUI that day, and the wall-clock seed (Renderers.cpp:1062-1063).  So the exposure is fitted (one
scalar, least squares in the log domain; all three files give 0.223 +- 0.001), the depth is the
one of {11, 12} named per file, and the comparison is statistical: the file must be as close to
a render of ours as a second, independently seeded render of ours is."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GAMMA = 2.1
WIDTH, HEIGHT = 768, 432
# (samples per pixel, path depth)
FILES = {5000: 11, 2500: 11, 128: 12}


def load(spp):
    path = os.path.join(HERE, "golden", "reference_renders", "MultLightGlass-0.4-%d-N.npz" % spp)
    rgb = np.load(path)["rgb"].astype(np.int32)
    assert rgb.shape == (HEIGHT, WIDTH, 3)
    return rgb


def file_order(colour_planes, spp):
    """(3, P) colour sums, bottom row first -> (H, W, 3) mean colour in the file's row order."""
    return (np.asarray(colour_planes, np.float64).reshape(3, HEIGHT, WIDTH)[:, ::-1]
            .transpose(1, 2, 0) / float(spp))


def fit_exposure(img, ref):
    ok = (ref > 8).all(axis=2) & (img > 1e-6).all(axis=2)
    return float(np.exp(np.mean(np.log(img[ok]) - GAMMA * np.log((ref[ok] + 0.5) / 255.999))))


def quantise(img, exposure):
    return np.floor(255.999 * (np.maximum(img, 0.0) / exposure) ** (1.0 / GAMMA)).astype(np.int32)


def pixel_stats(a, b):
    """Mean absolute difference in 8-bit levels; fraction of pixels within 8 levels in every channel."""
    d = np.abs(a - b)
    return float(d.mean()), float((d.max(axis=2) <= 8).mean())


def block_diff(a, b, k=16):
    """Mean absolute difference of k x k block means, in levels: per-pixel noise averages out
    (by a factor k), a bias does not."""
    hc, wc = HEIGHT // k * k, WIDTH // k * k
    ma = a[:hc, :wc].reshape(hc // k, k, wc // k, k, 3).mean(axis=(1, 3))
    mb = b[:hc, :wc].reshape(hc // k, k, wc // k, k, 3).mean(axis=(1, 3))
    return float(np.abs(ma - mb).mean())


def check(first, second, ref, what):
    """first, second: two independent renders (H, W, 3 mean colour, file row order)."""
    e = fit_exposure(first, ref)
    assert 0.218 < e < 0.228, (what, e)
    q0, q1 = quantise(first, e), quantise(second, e)
    d_ref, in8_ref = pixel_stats(q0, ref)
    d_self, in8_self = pixel_stats(q0, q1)
    b_ref, b_self = block_diff(q0, ref), block_diff(q0, q1)
    msg = "%s: exposure %.4f; per pixel %.3f vs %.3f levels, within 8 levels %.3f vs %.3f; " \
          "16x16 block means %.3f vs %.3f levels (file vs own second render)" % (
              what, e, d_ref, d_self, in8_ref, in8_self, b_ref, b_self)
    print(msg)
    assert d_ref <= 1.05 * d_self, msg                     # no farther than our own noise
    assert in8_ref >= in8_self - 0.015, msg
    assert b_ref <= 1.25 * b_self + 0.05, msg              # and no bias beyond it
    assert np.allclose(q0.mean(axis=(0, 1)), ref.mean(axis=(0, 1)), rtol=4e-3), msg
    return e, b_ref, b_self
