"""The tolerance-held form of the denoiser's feature pass (ptk_denoise variant 2) against the
CPU restatement and against variant 0.

Variant 2 keeps the discrete statistics (primaries, per-pixel deviations, the "median getter",
1/spp) bit for bit and forms the smooth ones a cheaper way: 7x7 mean and deviation from
per-column statistics shared through LDS, the Sobel magnitude from symmetric partial sums,
square roots from the hardware instruction.  Bars, as for variant 0: smooth features within
2e-5 relative (7x7 mean 2e-6) above a floor of a few ulp of the channel's magnitude,
bandwidths, filtered colour and weight sum within 2e-5 relative (5e-5 on fuzzed inputs)."""
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, PIN_SEED, scene_path, weights_path

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))

FAST = 2


@pytest.fixture(scope="module")
def ctx():
    import ptk
    c = ptk.Context(0)
    yield c
    c.close()


def _same(a, b):
    return (a == b) | (np.isnan(a) & np.isnan(b))


def _rel_ok(a, b, rel, abs_=1e-7):
    a = a.astype(np.float64)
    b = b.astype(np.float64)
    return _same(a, b) | (np.abs(a - b) <= rel * np.abs(b) + abs_)


@pytest.mark.parametrize("nwin", [1, 3])
def test_fast_features_match_golden(ctx, nwin, oracle):
    g = np.load(os.path.join(GOLDEN, "denoise_MultiLightGlass_128x72.npz"))
    O = oracle
    sc = O.load_scene(scene_path("MultiLightGlass"))
    ctx.set_constants(O.build_constants(sc, 128, 72, 4, 0.005, True))
    n = int(g["sample_count"])
    acc = g["acc_8spp"]
    ctx.write_planes(acc, n)
    w = O.load_weights(weights_path("BEST"))
    r = ctx.denoise(w, nwin, FAST, want_features=True)
    f, fr = r["features"], g["features36"]
    for lo, hi in ((0, 10), (30, 36)):                      # primaries, sd, "MAD", 1/spp: bit for bit
        assert np.array_equal(f[:, lo:hi], fr[:, lo:hi])
    floor = 4e-7 * max(1.0, float(np.abs(acc[0:16]).max()) / n)
    assert _rel_ok(f[:, 10:15], fr[:, 10:15], 2e-6, floor).all()   # 7x7 mean
    assert _rel_ok(f[:, 15:20], fr[:, 15:20], 2e-5, floor).all()   # 7x7 standard deviation
    assert _rel_ok(f[:, 20:25], fr[:, 20:25], 2e-5, floor).all()   # Sobel magnitude
    assert _rel_ok(f[:, 25:30], fr[:, 25:30], 2e-5, floor).all()   # 3x3 mean deviation
    assert _rel_ok(r["variances"], g["variances7"], 2e-5).all()
    assert _rel_ok(r["denoised"], g["denoised_n%d" % nwin], 2e-5).all()
    assert _rel_ok(r["wcSum"], g["wcsum_n%d" % nwin], 2e-5).all()


def test_fast_features_odd_sizes_and_edges(ctx, oracle):
    """Clamp-to-edge on an image that is not a multiple of the tile, every window width."""
    O = oracle
    W, H = 37, 23
    sc = O.load_scene(scene_path("cornell_crystal"))
    consts = O.build_constants(sc, W, H, 4, 0.005, True)
    acc = O.new_accumulators(W, H)
    n, _, _ = O.render(consts, acc, 5, 0, PIN_SEED, threads=4)
    w = O.load_weights(weights_path("box_grid"))
    _, var = O.forward_prop(acc, W, H, n, w)
    ctx.set_constants(consts)
    ctx.write_planes(acc, n)
    for nwin in (0, 1, 2, 5):
        den, wc = O.filter(acc, var, W, H, n, nwin)
        r = ctx.denoise(w, nwin, FAST)
        assert _rel_ok(r["variances"], var, 2e-5).all()
        assert _rel_ok(r["denoised"], den, 5e-5).all()
        assert _rel_ok(r["wcSum"], wc, 5e-5).all()
    import ptk
    with pytest.raises(ptk.PtkError, match="variant"):
        ctx.denoise(w, 1, 3)            # only variants 0, 1 and 2 exist


def test_fast_features_on_random_accumulators(oracle):
    import gpu_denoise_fuzz
    bad = gpu_denoise_fuzz.fuzz(30, 5, verbose=False, variant=FAST)
    assert not bad, "\n".join(bad)


@pytest.mark.parametrize("size", [(1920, 1080), (3840, 2160)])
def test_fast_features_full_size_against_variant_0_and_cpu_crops(ctx, oracle, size):
    """A rendered frame at full size: every pixel's bandwidths, filtered colour and weight sum
    within 2e-5 of variant 0's (whose own bar against the restatement is the same); crops
    recomputed on the CPU; row bands tile the whole-frame result bit for bit (a column's
    statistics do not depend on the tile that forms them)."""
    import ptk
    import torch
    import multigpu
    O = oracle
    W, H = size
    sc = O.load_scene(scene_path("MultiLightGlass"))
    ctx.set_constants(O.build_constants(sc, W, H, 4, 0.005, True))
    w = O.load_weights(weights_path("BEST"))
    ctx.render(16, ptk.SEED_COUNTER)
    base = ctx.denoise(w, 1, 0)
    fast = ctx.denoise(w, 1, FAST)
    for k in ("variances", "denoised", "wcSum"):
        assert _rel_ok(fast[k], base[k], 2e-5).all(), k
    den, wc = multigpu.device_denoised(ctx, torch.device("cuda", 0))
    den.zero_()
    wc.zero_()
    torch.cuda.synchronize()
    for r in range(8):
        y0, y1 = multigpu.row_band(H, 8, r)
        ctx.denoise_rows(w, 1, FAST, y0, y1)
    assert np.array_equal(den.cpu().numpy().T, fast["denoised"], equal_nan=True)
    assert np.array_equal(wc.cpu().numpy(), fast["wcSum"].reshape(-1), equal_nan=True)

    C = 48
    rng = np.random.default_rng(W)
    origins = [(0, 0), (W - C, H - C), (W // 2, 0)]
    origins += [(int(rng.integers(0, W - C)), int(rng.integers(0, H - C))) for _ in range(5)]
    checked = 0
    planes = ctx.read_planes().reshape(-1, H, W)
    for x0, y0 in origins:
        crop = np.ascontiguousarray(planes[:, y0:y0 + C, x0:x0 + C]).reshape(-1, C * C)
        _, var = O.forward_prop(crop, C, C, 16, w, threads=4)
        d, wsum = O.filter(crop, var, C, C, 16, 1, threads=4)
        ok = np.zeros((C, C), bool)
        ok[3 if y0 > 0 else 0:C - 3 if y0 + C < H else C, 3 if x0 > 0 else 0:C - 3 if x0 + C < W else C] = True
        sel = (slice(y0, y0 + C), slice(x0, x0 + C))
        assert np.allclose(fast["variances"].reshape(H, W, 7)[sel][ok], var.reshape(C, C, 7)[ok],
                           rtol=2e-5, atol=1e-7)
        assert np.allclose(fast["denoised"].reshape(H, W, 3)[sel][ok], d.reshape(C, C, 3)[ok],
                           rtol=2e-5, atol=1e-7, equal_nan=True)
        assert np.allclose(fast["wcSum"].reshape(H, W)[sel][ok], wsum.reshape(C, C)[ok],
                           rtol=2e-5, atol=1e-7)
        checked += int(ok.sum())
    assert checked > 12000
