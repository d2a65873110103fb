"""Statistics for comparing a render with an image the reference's authors rendered and
checked in (tests/golden/reference_renders/: Renders/MultLightGlass-0.4-<spp>-N.ppm).

What the file names and the reference's sources fix: the scene, 768x432 (resPerc 0.4 of
1920x1080, PT.cpp:53-54), the sample count, no denoising, gamma 2.1 (PT.cpp:72), randSamp 0.005
(PT.cpp:79), the post-process (c / exposure)^(1/g) and PT::SaveImage's int(255.999 c) without
clamping (PT.cpp:484-501).  The exposure is recorded too, by the scene file itself: its last
line is "pp 0.223 2.1" (Scenes/MultiLightGlass.scene:41; written by Scene::SaveScene from the
GUI's exposure and gamma, Scene.h:264, read back at Scene.h:213-218), and a least-squares fit
of that one scalar to each of the three images returns 0.2233 / 0.2228 / 0.2230: the absolute
scale of the radiance is pinned, not fitted away.  The images are therefore compared at the
recorded exposure, and the fit is only asserted to land on it.  What is not recorded: the path
depth set in the GUI that day and the wall-clock seed (Renderers.cpp:1062-1063).  The depth is
the one of {11, 12} named per file, and the comparison is statistical: the file must be as
close to a render of ours as a second, independently seeded render of ours is.

A second image, Renders/EdgeBlur.ppm (384x216), shows Scenes/cornell.scene (two boxes, one
rotated, under a box light): nothing about it is recorded, so the sample count is the one at
which our render carries the file's amount of pixel noise (36; the statistic `hf` below moves
3 % per 2 samples there), the depth is 12 (8 to 12 are indistinguishable at this noise level, 6
and less are off), and the exposure is the one fitted scalar.  scripts/
gpu_compare_reference_edgeblur.py holds the scan, profiles/r03_reference_edgeblur_scan.txt
its output.  The rest of that series (EdgeBlur0..4) went through denoiser settings that are
not recorded."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GAMMA = 2.1
RECORDED_EXPOSURE = 0.223       # "pp 0.223 2.1", the scene file's last line
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
    e_fit = fit_exposure(first, ref)
    assert abs(e_fit / RECORDED_EXPOSURE - 1.0) < 5e-3, (what, e_fit)   # the fit finds the recorded value
    e = RECORDED_EXPOSURE
    q0, q1 = quantise(first, e), quantise(second, e)
    d_ref, in8_ref = pixel_stats(q0, ref)
    d_self, in8_self = pixel_stats(q0, q1)
    b_ref, b_self = block_diff(q0, ref), block_diff(q0, q1)
    msg = "%s: exposure %.4f recorded, %.4f fitted; per pixel %.3f vs %.3f levels, within 8 levels " \
          "%.3f vs %.3f; 16x16 block means %.3f vs %.3f levels (file vs own second render)" % (
              what, e, e_fit, d_ref, d_self, in8_ref, in8_self, b_ref, b_self)
    print(msg)
    assert d_ref <= 1.05 * d_self, msg                     # no farther than our own noise
    assert in8_ref >= in8_self - 0.015, msg
    assert b_ref <= 1.25 * b_self + 0.05, msg              # and no bias beyond it
    assert np.allclose(q0.mean(axis=(0, 1)), ref.mean(axis=(0, 1)), rtol=4e-3), msg
    return e_fit, b_ref, b_self


# ---- Renders/EdgeBlur.ppm: Scenes/cornell.scene, 384 x 216
EDGEBLUR = {"scene": "cornell", "width": 384, "height": 216, "spp": 36, "depth": 12}


def load_edgeblur():
    rgb = np.load(os.path.join(HERE, "golden", "reference_renders", "EdgeBlur.npz"))["rgb"].astype(np.int32)
    assert rgb.shape == (EDGEBLUR["height"], EDGEBLUR["width"], 3)
    return rgb


def edgeblur_file_order(colour_planes, spp):
    h, w = EDGEBLUR["height"], EDGEBLUR["width"]
    return np.asarray(colour_planes, np.float64).reshape(3, h, w)[:, ::-1].transpose(1, 2, 0) / float(spp)


def _blocks(a, k=16):
    h, w = a.shape[0] // k * k, a.shape[1] // k * k
    return a[:h, :w].reshape(h // k, k, w // k, k, 3).mean(axis=(1, 3))


def block_diff_any(a, b, k=16):
    return float(np.abs(_blocks(a.astype(np.float64), k) - _blocks(b.astype(np.float64), k)).mean())


def fine_structure(img):
    """RMS of (image - its 5x5 box mean), in levels: at a few dozen samples per pixel this is
    the pixel noise, so it tells the sample count."""
    from scipy.ndimage import uniform_filter
    img = img.astype(np.float64)
    return float(np.sqrt(np.mean((img - uniform_filter(img, size=(5, 5, 1), mode="nearest")) ** 2)))


def fit_exposure_blocks(img, ref):
    """The scalar that brings the block means of the quantised image closest to the file's.
    Both sides are then means of gamma-mapped noisy pixels, so the downward pull that noise
    exerts on such a mean is the same when the sample counts match."""
    lin_ref = ((ref + 0.5) / 255.999) ** GAMMA
    e0 = float(img.mean() / lin_ref.mean())
    cands = [e0 * (0.90 + 0.005 * i) for i in range(41)]
    return min(cands, key=lambda c: block_diff_any(quantise(img, c), ref))


def check_edgeblur(first, second, what):
    """first, second: two independent renders of the cornell scene (H, W, 3 mean colour, file
    row order) at EDGEBLUR's size, sample count and depth."""
    ref = load_edgeblur()
    e = fit_exposure_blocks(first, ref)
    q0, q1 = quantise(first, e), quantise(second, e)
    d_ref, d_self = float(np.abs(q0 - ref).mean()), float(np.abs(q0 - q1).mean())
    b_ref, b_self = block_diff_any(q0, ref), block_diff_any(q0, q1)
    f_own, f_ref = fine_structure(q0), fine_structure(ref)
    msg = "%s: exposure %.4f; per pixel %.3f vs %.3f levels; 16x16 block means %.3f vs %.3f levels " \
          "(file vs own second render); fine structure %.3f vs %.3f (ours vs file)" % (
              what, e, d_ref, d_self, b_ref, b_self, f_own, f_ref)
    print(msg)
    assert d_ref <= 1.03 * d_self, msg                     # no farther than our own noise
    assert b_ref <= 1.15 * b_self + 0.05, msg              # no bias beyond it
    assert abs(f_own / f_ref - 1.0) <= 0.03, msg           # the same amount of noise
    assert np.allclose(q0.mean(axis=(0, 1)), ref.mean(axis=(0, 1)), rtol=5e-3), msg
    return e, b_ref, b_self
