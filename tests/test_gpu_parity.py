"""GPU parity: the HIP path (through the C ABI) against the CPU oracle and the
committed golden fixtures.

Tolerances (PTK_MATH_EXACT, matched reference seeds).  The kernel performs the
reference's fp32 operations in the reference's order with IEEE divide/sqrt and
no FMA contraction, and evaluates cos/sin/acos and the Schlick power in double,
rounding once.  The oracle has two builds to compare with:

  * variant R (cos/sin/acos/pow correctly rounded): the kernel must equal it BIT
    FOR BIT -- every pixel, every sample, all 38 planes, NaN for NaN.
  * variant F (glibc's cosf/sinf/acosf/powf, faithfully but not correctly
    rounded; the canonical build): a one-ulp difference at one of those calls
    perturbs a path's colour by ~1e-7 relative and on rare paths flips a
    discrete decision (closest shape, refract/reflect, shadow test), after
    which that sample is a different, equally valid path.  Against F: >= 99 %
    of pixels bit-identical in colour at 1 spp, >= 99.9 % within 1e-4
    relative; 8 spp accumulators >= 99.5 % within 1e-4 and the image mean
    within 1e-4 (the libm-variant noise floor of the reference itself is 6e-5).
  * RNG streams, ray counts, seed state: bit-exact against both.
  * exposure: raw sum within 1e-9 relative of the oracle's double sum, scaled
    value within 2e-6 relative of the oracle's sequential-fp32 value at these
    sizes.
  * denoiser: discrete feature planes bit-exact; smooth planes, bandwidths,
    filtered colour and weight sum within 2e-5 relative.
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN, PIN_SEED, scene_path, weights_path

pytestmark = pytest.mark.gpu

SCENES = ["cornell", "cornell_crystal", "MultiLightGlass", "Box_Grid", "blur_test", "bars3"]


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


@pytest.mark.parametrize("name", SCENES)
def test_one_sample_per_pixel_matches_golden(ctx, name):
    import ptk
    g = np.load(os.path.join(GOLDEN, "render_%s_64x36.npz" % name))
    ctx.set_math_mode(ptk.MATH_EXACT)
    ctx.set_constants(g["constants"].tobytes())
    rays, glo = ctx.render(1, ptk.SEED_REFERENCE, PIN_SEED)
    planes = ctx.read_planes()
    ref = g["samples_1spp"]            # (P, 17): xyz, normal, alb1, alb2, wpos, dl, rays
    assert rays == int(ref[:, 16].sum())
    col_same = _same(planes[0:3].T, ref[:, 0:3]).all(axis=1)
    assert col_same.mean() >= 0.99
    assert _rel_ok(planes[0:3].T, ref[:, 0:3], 1e-4).all(axis=1).mean() >= 0.999
    feat_same = _same(planes[3:16].T, ref[:, 3:16]).all(axis=1)
    assert feat_same.mean() >= 0.999
    # running statistics after one sample: deviation of the sample from itself
    assert np.all(planes[32:38][:, col_same] == 0.0)


@pytest.mark.parametrize("name", SCENES)
def test_eight_samples_match_golden_accumulators(ctx, name, oracle):
    import ptk
    g = np.load(os.path.join(GOLDEN, "render_%s_64x36.npz" % name))
    ctx.set_math_mode(ptk.MATH_EXACT)
    ctx.set_constants(g["constants"].tobytes())
    # two calls (3 + 5) must equal one call of 8: the statistics carry over
    _, glo = ctx.render(3, ptk.SEED_REFERENCE, PIN_SEED)
    _, glo = ctx.render(5, ptk.SEED_REFERENCE, glo)
    assert glo == tuple(int(x) for x in g["glo_after"])
    assert ctx.sample_count == 8
    planes = ctx.read_planes()
    ref = g["acc_8spp"]
    ok = _rel_ok(planes, ref, 1e-4).all(axis=0)
    assert ok.mean() >= 0.995
    m_gpu = np.nanmean(planes[0:3].astype(np.float64), axis=1)
    m_ref = np.nanmean(ref[0:3].astype(np.float64), axis=1)
    assert np.all(np.abs(m_gpu / m_ref - 1.0) < 1e-4)
    # exposure on the GPU's own accumulators vs the oracle on the same planes
    for f in (0, 1, 2):
        e_gpu, raw_gpu = ctx.auto_exposure(f)
        e_ref, raw_ref = oracle.auto_exposure(planes, 64, 36, 8, f)
        assert abs(raw_gpu - raw_ref) <= 1e-9 * abs(raw_ref)
        assert abs(e_gpu - e_ref) <= 2e-6 * abs(e_ref)


def _mismatch_report(got, ref, W):
    bad = ~_same(got, ref)
    if not bad.any():
        return ""
    planes, pixels = np.nonzero(bad)
    first = ["plane %d pixel (row %d, col %d): %r vs %r" % (p, i // W, i % W, got[p, i], ref[p, i])
             for p, i in list(zip(planes, pixels))[:8]]
    return "%d values in %d pixels differ; first: %s" % (bad.sum(), len(set(pixels)),
                                                         "; ".join(first))


@pytest.mark.parametrize("name", SCENES)
def test_exact_mode_equals_the_rounded_libm_oracle_bit_for_bit(ctx, name):
    """Against variant R of the oracle (cos/sin/acos/pow correctly rounded) nothing may
    differ: one sample, then eight, all planes of every pixel."""
    import ptk
    g = np.load(os.path.join(GOLDEN, "render_%s_64x36.npz" % name))
    ctx.set_math_mode(ptk.MATH_EXACT)
    ctx.set_constants(g["constants"].tobytes())
    rays, glo = ctx.render(1, ptk.SEED_REFERENCE, PIN_SEED)
    ref = g["samples_1spp_R"]
    assert rays == int(ref[:, 16].sum())
    assert _mismatch_report(ctx.read_planes(0, 16), np.ascontiguousarray(ref[:, 0:16].T), 64) == ""
    rays7, glo = ctx.render(7, ptk.SEED_REFERENCE, glo)
    assert glo == tuple(int(x) for x in g["glo_after"])
    assert _mismatch_report(ctx.read_planes(), g["acc_8spp_R"], 64) == ""


@pytest.mark.parametrize("name", ["MultiLightGlass", "Box_Grid", "cornell_crystal"])
def test_one_long_launch_equals_launches_of_every_length(ctx, name):
    """The feature sums are brought up to date every few samples inside a launch, from
    per-sample records.  27 samples in one launch (three full batches and a remainder) must
    leave every plane with the bits that 27 launches of one sample leave, and that launches of
    8, 1, 9, 7 and 2 leave; the seed state and ray counts likewise."""
    import ptk
    g = np.load(os.path.join(GOLDEN, "render_%s_64x36.npz" % name))
    ctx.set_math_mode(ptk.MATH_EXACT)
    ctx.set_constants(g["constants"].tobytes())
    rays_one, glo_one = ctx.render(27, ptk.SEED_REFERENCE, PIN_SEED)
    one = ctx.read_planes()
    for split in ([1] * 27, [8, 1, 9, 7, 2]):
        ctx.reset()
        glo, rays = PIN_SEED, 0
        for n in split:
            r, glo = ctx.render(n, ptk.SEED_REFERENCE, glo)
            rays += r
        assert ctx.sample_count == 27
        assert glo == glo_one and rays == rays_one
        assert _mismatch_report(ctx.read_planes(), one, 64) == ""


def test_read_accumulators_uses_the_reference_layout(ctx):
    import ptk
    g = np.load(os.path.join(GOLDEN, "render_cornell_64x36.npz"))
    ctx.set_constants(g["constants"].tobytes())
    ctx.write_planes(g["acc_8spp"], 8)
    acc = ctx.read_accumulators()
    assert acc["sampleCount"] == 8
    ref = g["acc_8spp"]
    assert np.array_equal(acc["preScreen"], ref[0:3].T)
    assert np.array_equal(acc["normal"], ref[3:6].T)
    assert np.array_equal(acc["albedo1"], ref[6:9].T)
    assert np.array_equal(acc["albedo2"], ref[9:12].T)
    assert np.array_equal(acc["worldPos"], ref[12:15].T)
    assert np.array_equal(acc["directLight"], ref[15])
    assert np.array_equal(acc["stdDev"], ref[32:38].T)


@pytest.mark.parametrize("nwin", [1, 3])
def test_denoiser_matches_golden(ctx, nwin, oracle, variant=0):
    g = np.load(os.path.join(GOLDEN, "denoise_MultiLightGlass_128x72.npz"))
    O = oracle
    sc = O.load_scene(scene_path("MultiLightGlass"))
    ctx.set_constants(O.build_constants(sc, 128, 72, 4, 0.005, True))
    n = int(g["sample_count"])
    ctx.write_planes(g["acc_8spp"], n)
    w = O.load_weights(weights_path("BEST"))
    r = ctx.denoise(w, nwin, variant, want_features=(variant == 0))
    if variant == 0:
        f, fr = r["features"], g["features36"]
        for lo, hi in ((0, 15), (20, 25), (30, 36)):      # primaries, sd, mean7, Sobel, MAD, 1/spp
            assert np.array_equal(f[:, lo:hi], fr[:, lo:hi])
        assert _rel_ok(f[:, 15:20], fr[:, 15:20], 2e-5).all()  # 7x7 standard deviation
        assert _rel_ok(f[:, 25:30], fr[:, 25:30], 2e-5).all()  # 3x3 mean deviation
    assert _rel_ok(r["variances"], g["variances7"], 2e-5).all()
    assert _rel_ok(r["denoised"], g["denoised_n%d" % nwin], 2e-5).all()
    assert _rel_ok(r["wcSum"], g["wcsum_n%d" % nwin], 2e-5).all()


def test_denoiser_odd_sizes_and_edges(ctx, oracle):
    """Clamp-to-edge on an image that is not a multiple of the 16x16 tile."""
    O = oracle
    W, H = 37, 23
    sc = O.load_scene(scene_path("cornell_crystal"))
    consts = O.build_constants(sc, W, H, 4, 0.005, True)
    acc = O.new_accumulators(W, H)
    n, _, _ = O.render(consts, acc, 5, 0, PIN_SEED, threads=4)
    w = O.load_weights(weights_path("box_grid"))
    feats, var = O.forward_prop(acc, W, H, n, w)
    ctx.set_constants(consts)
    ctx.write_planes(acc, n)
    for nwin in (0, 1, 2, 5):
        den, wc = O.filter(acc, var, W, H, n, nwin)
        r = ctx.denoise(w, nwin, 0)
        assert _rel_ok(r["variances"], var, 2e-5).all()
        assert _rel_ok(r["denoised"], den, 5e-5).all()
        assert _rel_ok(r["wcSum"], wc, 5e-5).all()
    import ptk
    with pytest.raises(ptk.PtkError, match="variant"):
        ctx.denoise(w, 1, 3)            # only variants 0, 1 and 2 exist


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(37, 23), (61, 3), (62, 8), (63, 9), (64, 17), (253, 131), (1920, 1080)])
def test_filter_row_window_kernel_equals_the_tile_kernel_bit_for_bit(ctx, oracle, size):
    """N = 1 runs on a register window of rows (62-column strips, walked down or up the
    image, neighbours through DPP operands); variant 1 forces the LDS-tile kernel the wider
    windows use.  Both evaluate the same taps in the same order with the same roundings, so
    every output word must be identical: whole image and ragged row bands, at widths around
    the strip and wave sizes and at full size."""
    O = oracle
    W, H = size
    rng = np.random.default_rng(W * 1000 + H)
    P = W * H
    spp = 7                                             # not a power of two: exact quotients
    pl = np.zeros((38, P), np.float32)
    pl[0:16] = (rng.uniform(-1, 1, (16, P)) * spp).astype(np.float32)
    pl[15] = rng.choice([0, .25, .5, .75, 1.0], P) * spp
    pl[32:38] = rng.uniform(1e-4, 0.5, (6, P))
    sc = O.load_scene(scene_path("cornell_crystal"))
    ctx.set_constants(O.build_constants(sc, W, H, 4, 0.005, True))
    ctx.write_planes(pl, spp)
    w = O.load_weights(weights_path("box_grid"))
    a = ctx.denoise(w, 1, 0)
    b = ctx.denoise(w, 1, 1)
    for k in ("denoised", "wcSum", "variances"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
    if H >= 9:                                          # a band that starts and ends mid-image
        import torch
        import multigpu
        den, wc = multigpu.device_denoised(ctx, torch.device("cuda", 0))
        def band(variant, y0, y1):
            den.zero_()
            wc.zero_()
            torch.cuda.synchronize()
            ctx.denoise_rows(w, 1, variant, y0, y1)
            ctx.synchronize()
            return np.concatenate([den.cpu().numpy().reshape(3, P), wc.cpu().numpy().reshape(1, P)])
        y0, y1 = 2, H - 3
        b0, b1 = band(0, y0, y1), band(1, y0, y1)
        assert np.array_equal(b0.view(np.uint32), b1.view(np.uint32))
        assert not b0[:, :y0 * W].any() and not b0[:, y1 * W:].any()    # nothing outside the band
        full = np.concatenate([a["denoised"].T, a["wcSum"].reshape(1, P)])
        assert np.array_equal(b0[:, y0 * W:y1 * W].view(np.uint32), full[:, y0 * W:y1 * W].view(np.uint32))


def test_render_odd_sizes_depths_and_empty_scene(ctx, oracle):
    import ptk
    O = oracle
    sc = O.load_scene(scene_path("bars3"))
    for (W, H, depth) in ((1, 1, 4), (17, 9, 1), (33, 20, 7), (40, 30, 12)):
        consts = O.build_constants(sc, W, H, depth, 0.005, True)
        acc = O.new_accumulators(W, H)
        n, glo, rays = O.render(consts, acc, 2, 0, PIN_SEED, threads=2)
        ctx.set_constants(consts)
        _, gglo = ctx.render(2, ptk.SEED_REFERENCE, PIN_SEED)
        assert gglo == glo
        planes = ctx.read_planes()
        assert _rel_ok(planes, acc, 1e-4).all(axis=0).mean() >= 0.98
        with O.variant("R"):
            acc_r = O.new_accumulators(W, H)
            O.render(consts, acc_r, 2, 0, PIN_SEED, threads=2)
        assert _mismatch_report(planes, acc_r, W) == ""
    # no shapes: every ray misses and sees the sky gradient
    consts = O.build_constants(sc, 32, 16, 4, 0.0, False)
    consts.num_shapes = 0
    consts.num_imp = 0
    acc = O.new_accumulators(32, 16)
    O.render(consts, acc, 1, 0, PIN_SEED)
    ctx.set_constants(consts)
    rays, _ = ctx.render(1, ptk.SEED_REFERENCE, PIN_SEED)
    assert rays == 32 * 16
    assert np.array_equal(ctx.read_planes(0, 3), acc[0:3])


def test_invalid_arguments_are_rejected(ctx, oracle):
    import ptk
    O = oracle
    sc = O.load_scene(scene_path("cornell"))
    consts = O.build_constants(sc, 32, 16, 4, 0.005, True)
    with pytest.raises(ptk.PtkError):
        ctx.set_constants(bytes(consts)[:-4])
    bad = O.build_constants(sc, 32, 16, 4, 0.005, True)
    bad.num_shapes = 51
    with pytest.raises(ptk.PtkError):
        ctx.set_constants(bad)
    bad = O.build_constants(sc, 32, 16, 13, 0.005, True)
    with pytest.raises(ptk.PtkError):
        ctx.set_constants(bad)
    bad = O.build_constants(sc, 32, 16, 4, 0.005, True)
    bad.shapes[2][2] = 445          # 13 attributes would run past the 450-entry table
    with pytest.raises(ptk.PtkError):
        ctx.set_constants(bad)
    ctx.set_constants(consts)
    with pytest.raises(ptk.PtkError):
        ctx.render(-1, ptk.SEED_COUNTER)
    with pytest.raises(ptk.PtkError):
        ctx.denoise(O.load_weights(weights_path("BEST")), 1, 0)   # nothing accumulated yet
    rays, _ = ctx.render(0, ptk.SEED_COUNTER)
    assert rays == 0 and ctx.sample_count == 0


def test_counter_seeds_are_deterministic_and_additive(ctx, oracle):
    """Full-size properties at 1080p: a call split in two equals one call, a repeat
    reproduces it bit for bit, and different streams decorrelate."""
    import ptk
    O = oracle
    W, H = 1920, 1080
    sc = O.load_scene(scene_path("cornell_crystal"))
    ctx.set_constants(O.build_constants(sc, W, H, 4, 0.005, True))
    ctx.set_stream_id(0)
    ctx.render(6, ptk.SEED_COUNTER)
    a = ctx.read_planes()
    ctx.reset()
    ctx.render(2, ptk.SEED_COUNTER)
    ctx.render(4, ptk.SEED_COUNTER)
    b = ctx.read_planes()
    assert np.array_equal(a, b, equal_nan=True)
    ctx.reset()
    ctx.set_stream_id(1)
    ctx.render(6, ptk.SEED_COUNTER)
    c = ctx.read_planes(0, 3)
    ctx.set_stream_id(0)
    assert not np.array_equal(a[0:3], c)
    ma = np.nanmean(a[0:3].astype(np.float64)) / 6
    mc = np.nanmean(c.astype(np.float64)) / 6
    assert abs(ma / mc - 1.0) < 5e-3     # two independent 12 M-sample estimates of one mean
    e, raw = ctx.auto_exposure(2)
    assert raw > 0 and np.isfinite(e)


@pytest.mark.parametrize("name", ["cornell_crystal", "MultiLightGlass"])
def test_full_size_frame_spot_checked_against_the_cpu(ctx, oracle, name):
    """1920x1080 with the reference's sequential seed stream (33 MB of seeds per pass): two
    passes on the GPU, then 4000 pixels drawn at random are recomputed on the CPU from the
    same seeds.  Same bar as the small frames: >= 99 % of them bit-identical per sample."""
    import ptk
    O = oracle
    W, H = 1920, 1080
    P = W * H
    consts = O.build_constants(O.load_scene(scene_path(name)), W, H, 4, 0.005, True)
    ctx.set_constants(consts)
    rays, glo = ctx.render(1, ptk.SEED_REFERENCE, PIN_SEED)
    first = ctx.read_planes(0, 16)
    seeds1, glo1 = O.seed_stream(PIN_SEED, P)
    assert glo == glo1
    rays2, glo2 = ctx.render(1, ptk.SEED_REFERENCE, glo)
    both = ctx.read_planes(0, 16)
    seeds2, glo_end = O.seed_stream(glo1, P)
    assert glo2 == glo_end
    rng = np.random.default_rng(5)
    picks = rng.choice(P, 4000, replace=False)

    def recompute():
        same1 = same2 = 0
        for i in picks:
            row, col = divmod(int(i), W)
            s = O.render_sample(consts, row, col, int(seeds1[i, 0]), int(seeds1[i, 1]))
            ref1 = np.float32(list(s.xyz) + list(s.normal) + list(s.albedo1) + list(s.albedo2)
                              + list(s.world_pos) + [s.direct_light])
            ok1 = np.array_equal(first[:, i], ref1, equal_nan=True)
            same1 += ok1
            s = O.render_sample(consts, row, col, int(seeds2[i, 0]), int(seeds2[i, 1]))
            ref2 = np.float32(list(s.xyz) + list(s.normal) + list(s.albedo1) + list(s.albedo2)
                              + list(s.world_pos) + [s.direct_light])
            same2 += ok1 and np.array_equal(both[:, i], ref1 + ref2, equal_nan=True)
        return same1, same2

    same1, same2 = recompute()            # canonical variant F: libm-limited
    assert same1 >= 0.99 * len(picks)
    assert same2 >= 0.98 * len(picks)
    with O.variant("R"):                  # correctly rounded libm: nothing may differ
        same1, same2 = recompute()
    assert same1 == len(picks) and same2 == len(picks)


@pytest.mark.parametrize("name,mse_tol", [("cornell_crystal", 0.05), ("MultiLightGlass", 0.05),
                                          ("Box_Grid", 0.10)])
def test_counter_seeds_are_statistically_equivalent_to_reference_seeds(ctx, name, mse_tol):
    """The device's counter-based seeds (the benchmarked mode) against the reference's
    sequential seed stream, both in exact math: 256 samples of each, measured against one
    4096-spp render on a third stream.  The two 256-spp images must be equally good estimates:
    relative MSE within 5 % of each other (10 % on Box_Grid, whose error is dominated by few
    bright pixels), image means within 0.2 % of the 4096-spp mean."""
    import ptk
    W, H = 240, 135
    raw, _, _ = ptk.scene_constants(scene_path(name), W, H, 4, 0.005, False)
    ctx.set_math_mode(ptk.MATH_EXACT)
    ctx.set_constants(raw)
    ctx.set_stream_id(1234)
    ctx.render(4096, ptk.SEED_COUNTER)
    ref = ctx.read_planes(0, 3).astype(np.float64) / 4096.0
    ctx.set_stream_id(0)

    def quality(img):
        return np.nanmean(img), np.nanmean((img - ref) ** 2 / (ref ** 2 + 1e-3))

    ctx.reset()
    ctx.render(256, ptk.SEED_COUNTER)
    m_ctr, mse_ctr = quality(ctx.read_planes(0, 3).astype(np.float64) / 256.0)
    ctx.reset()
    ctx.render(256, ptk.SEED_REFERENCE, PIN_SEED)
    m_ref, mse_ref = quality(ctx.read_planes(0, 3).astype(np.float64) / 256.0)
    m = np.nanmean(ref)
    assert abs(m_ctr / m - 1.0) < 2e-3
    assert abs(m_ref / m - 1.0) < 2e-3
    assert abs(mse_ctr / mse_ref - 1.0) < mse_tol


def test_counter_seeds_are_uncorrelated_across_pixels_and_samples(ctx, oracle):
    """Independence of the per-(pixel, sample, stream) seeds, observed through the kernel:
    with no shapes every path is the camera ray into the sky, whose blue channel is a smooth
    monotone function of the second jitter draw.  Residuals about each pixel's own mean must
    be uncorrelated between neighbouring pixels (same sample), between consecutive samples
    (same pixel) and between streams, within three standard errors of zero."""
    import ptk
    O = oracle
    W, H, S = 160, 90, 48
    consts = O.build_constants(O.load_scene(scene_path("cornell")), W, H, 1, 0.5, False)
    consts.num_shapes = 0
    consts.num_imp = 0

    def residuals(stream):
        ctx.set_constants(consts)
        ctx.set_stream_id(stream)
        v = np.empty((S, H, W), np.float64)
        prev = np.zeros(W * H, np.float64)
        for k in range(S):
            ctx.render(1, ptk.SEED_COUNTER)
            acc = ctx.read_planes(2, 1)[0].astype(np.float64)
            v[k] = (acc - prev).reshape(H, W)
            prev = acc
        return v - v.mean(axis=0, keepdims=True)

    r = residuals(0)
    assert r.std() > 0.01                       # the jitter does move the ray

    def corr(a, b):
        return float((a * b).sum() / np.sqrt((a * a).sum() * (b * b).sum()))

    # removing each pixel's mean over S samples leaves a correlation of -1/(S-1) between
    # samples of one pixel; it does not touch the correlation between different pixels
    checks = {
        "next pixel in the row": (corr(r[:, :, :-1], r[:, :, 1:]), 0.0, S * H * (W - 1)),
        "next pixel in the column": (corr(r[:, :-1, :], r[:, 1:, :]), 0.0, S * (H - 1) * W),
        "next sample": (corr(r[:-1], r[1:]), -1.0 / (S - 1), (S - 1) * H * W),
        "next stream": (corr(r, residuals(1)), 0.0, S * H * W),
    }
    ctx.set_stream_id(0)
    for what, (c, expect, n) in checks.items():
        assert abs(c - expect) < 3.0 / np.sqrt(n), (what, c, expect, n)


def test_denoise_row_bands_tile_the_full_result(ctx, oracle):
    """Bands of any height (not tile aligned) reproduce the whole-image denoise bit for bit."""
    import ptk
    import torch
    import multigpu
    O = oracle
    W, H = 157, 83
    sc = O.load_scene(scene_path("MultiLightGlass"))
    ctx.set_constants(O.build_constants(sc, W, H, 4, 0.005, True))
    w = O.load_weights(weights_path("BEST"))
    ctx.render(16, ptk.SEED_COUNTER)
    for variant in (0,):
        full = ctx.denoise(w, 2, variant)
        den, wc = multigpu.device_denoised(ctx, torch.device("cuda", 0))
        den.zero_()
        wc.zero_()
        torch.cuda.synchronize()
        for y0, y1 in ((0, 5), (5, 5), (5, 41), (41, 82), (82, 83)):
            ctx.denoise_rows(w, 2, variant, y0, y1)
        got = den.cpu().numpy().T
        assert np.array_equal(got, full["denoised"], equal_nan=True)
        assert np.array_equal(wc.cpu().numpy(), full["wcSum"].reshape(-1), equal_nan=True)
    with pytest.raises(ptk.PtkError):
        ctx.denoise_rows(w, 1, 0, 10, 5)
    with pytest.raises(ptk.PtkError):
        ctx.denoise_rows(w, 1, 0, 0, H + 1)


def test_full_size_denoise_properties(ctx, oracle):
    """1920x1080, 16 spp with features, BEST.txt: the eight row bands of an 8-GPU split tile
    the whole-frame result bit for bit; the bandwidths of twelve random 48x48 crops recomputed on the CPU from the read-back
    accumulators agree within the small-frame tolerance."""
    import ptk
    import torch
    import multigpu
    O = oracle
    W, H = 1920, 1080
    sc = O.load_scene(scene_path("MultiLightGlass"))
    ctx.set_constants(O.build_constants(sc, W, H, 4, 0.005, True))
    w = O.load_weights(weights_path("BEST"))
    ctx.render(16, ptk.SEED_COUNTER)
    full = ctx.denoise(w, 1, 0)
    assert np.isfinite(full["denoised"]).mean() > 0.999
    den, wc = multigpu.device_denoised(ctx, torch.device("cuda", 0))
    den.zero_()
    wc.zero_()
    torch.cuda.synchronize()
    for r in range(8):
        y0, y1 = multigpu.row_band(H, 8, r)
        ctx.denoise_rows(w, 1, 0, y0, y1)
    assert np.array_equal(den.cpu().numpy().T, full["denoised"], equal_nan=True)
    # bandwidths of random pixels against the CPU, computed from a crop around each
    planes = ctx.read_planes()
    rng = np.random.default_rng(9)
    var_gpu = full["variances"]
    checked = 0
    for _ in range(12):
        x0, y0 = int(rng.integers(0, W - 48)), int(rng.integers(0, H - 48))
        crop = np.ascontiguousarray(planes.reshape(-1, H, W)[:, y0:y0 + 48, x0:x0 + 48]).reshape(-1, 48 * 48)
        _, var = O.forward_prop(crop, 48, 48, 16, w, threads=4)
        inner = np.zeros((48, 48), bool)
        inner[3:45, 3:45] = True              # the crop's own edges clamp differently
        g = var_gpu.reshape(H, W, 7)[y0:y0 + 48, x0:x0 + 48][inner]
        assert np.allclose(g, var.reshape(48, 48, 7)[inner], rtol=2e-5, atol=1e-7)
        checked += int(inner.sum())
    assert checked > 20000


def test_post_process_matches_the_display_transform(ctx, oracle):
    """Display metrics 0..7: float screen within 2e-6 relative of the CPU restatement (the
    gamma power is exp2(log2(v) / g) on the hardware units here, libm's powf there); 8-bit
    output identical except where a value sits within that distance of a quantisation step."""
    import ptk
    O = oracle
    W, H = 96, 54
    sc = O.load_scene(scene_path("MultiLightGlass"))
    ctx.set_constants(O.build_constants(sc, W, H, 4, 0.005, True))
    ctx.render(6, ptk.SEED_COUNTER)
    planes = ctx.read_planes()
    w = O.load_weights(weights_path("BEST"))
    with pytest.raises(ptk.PtkError):
        ctx.post_process(1, 1.0, 2.2)                      # nothing denoised yet
    den = ctx.denoise(w, 1, 0)["denoised"]
    for metric, exposure, g in ((0, 1.0, 2.2), (0, 0.37, 1.0), (1, 1.3, 2.2), (2, 1.0, 2.2),
                                (3, 1.0, 1.7), (4, 2.0, 2.2), (5, 1.0, 2.2), (6, 1.0, 2.2)):
        ref = O.post_process(planes, den, W, H, 6, metric, exposure, g)
        post, rgb = ctx.post_process(metric, exposure, g, want_rgb8=True)
        assert _rel_ok(post, ref, 2e-6).all(), metric
        ref8 = O.quantise8(ref, W, H)
        assert (rgb != ref8).mean() < 1e-3
        assert np.abs(rgb.astype(int) - ref8.astype(int)).max() <= 1
        # the library's own float screen quantises to exactly its 8-bit output
        assert np.array_equal(rgb, O.quantise8(post, W, H))
    # metric 7: the training target (CUDARender.cu:1006-1010), once one is held
    with pytest.raises(ptk.PtkError, match="metric 7"):
        ctx.post_process(7, 1.0, 2.2)
    ctx.save_target()
    target = np.ascontiguousarray(planes[0:3].T / np.float32(6))
    for exposure, g in ((1.0, 2.2), (0.6, 1.8)):
        ref = O.post_process(planes, target, W, H, 6, 7, exposure, g)
        post, rgb = ctx.post_process(7, exposure, g, want_rgb8=True)
        assert _rel_ok(post, ref, 2e-6).all()
        assert np.array_equal(rgb, O.quantise8(post, W, H))
    custom = np.random.default_rng(3).random((W * H, 3), dtype=np.float32)
    ctx.set_target(custom)
    post, _ = ctx.post_process(7, 1.0, 1.0)
    assert _rel_ok(post, custom, 2e-6).all()
    with pytest.raises(ptk.PtkError):
        ctx.post_process(8, 1.0, 2.2)
    assert ctx.timing().post_ms > 0.0


@pytest.mark.parametrize("name,mean_tol", [("cornell_crystal", 2e-3), ("MultiLightGlass", 2e-3),
                                           ("cornell", 2e-3), ("Box_Grid", 5e-3)])
def test_fast_math_is_statistically_equivalent(ctx, name, mean_tol):
    """PTK_MATH_FAST (reciprocal slabs, rsq, hardware sin/cos) follows different paths
    sample by sample, so it is held to the RMSE criterion: against a 4096-spp exact render,
    256 fast samples have the exact mode's relative MSE (within 5 %; 10 % on Box_Grid) and the
    image mean agrees within 0.2 % -- except on Box_Grid, where the reciprocal slab quotients
    shift hit points enough to change how often grazing rays re-hit their own surface: +0.14 %
    on the mean (bounded here at 0.5 %), which is why fast mode is not the default and not
    the benchmarked mode."""
    import ptk
    W, H = 240, 135
    raw, _, _ = ptk.scene_constants(scene_path(name), W, H, 4, 0.005, False)
    ctx.set_constants(raw)
    ctx.set_math_mode(ptk.MATH_EXACT)
    ctx.set_stream_id(77)
    ctx.render(4096, ptk.SEED_COUNTER)
    ref = ctx.read_planes(0, 3).astype(np.float64) / 4096.0
    out = {}
    for mode in (ptk.MATH_EXACT, ptk.MATH_FAST):
        ctx.set_constants(raw)
        ctx.set_math_mode(mode)
        ctx.set_stream_id(0)
        ctx.render(256, ptk.SEED_COUNTER)
        img = ctx.read_planes(0, 3).astype(np.float64) / 256.0
        out[mode] = (np.nanmean(img), np.nanmean((img - ref) ** 2 / (ref ** 2 + 1e-3)))
    ctx.set_math_mode(ptk.MATH_EXACT)
    ctx.set_stream_id(0)
    m_ref = np.nanmean(ref)
    assert abs(out[ptk.MATH_EXACT][0] / m_ref - 1.0) < 2e-3
    assert abs(out[ptk.MATH_FAST][0] / m_ref - 1.0) < mean_tol
    mse_tol = 0.10 if name == "Box_Grid" else 0.05
    assert abs(out[ptk.MATH_FAST][1] / out[ptk.MATH_EXACT][1] - 1.0) < mse_tol


def test_full_tables_fifty_shapes_fifty_materials(ctx, oracle):
    """The POD's limits (headers/GLOBALS.h:87-89): 50 shapes (25 boxes, most rotated, some flat,
    and 25 spheres: 425 of the 450 attributes), 50 materials of all four kinds, some blurred,
    10 important shapes, depth 12.  Same bars as the scene files after two samples with the
    reference seeds: >= 98 % of pixels within 1e-4 of variant F, every plane of every pixel
    bit-identical to variant R."""
    import ptk
    O = oracle
    rng = np.random.default_rng(2024)
    sc = O.load_scene(scene_path("cornell"))
    c = O.build_constants(sc, 48, 32, 12, 0.005, True)
    for m in range(50):
        kind = m % 4
        col = rng.uniform(0.2, 1.0, 3) * (4.0 if kind == 1 else 1.0)
        blur = 0.05 if (kind >= 2 and m % 3 == 0) else 0.0
        for k, v in enumerate((*col, blur, 1.5, float(kind))):
            c.mats[m][k] = v
    off = 0
    for s in range(50):
        pos = rng.uniform(-6, 6, 3) + np.array([0.0, 0.0, 14.0])
        if s < 25:
            half = rng.uniform(0.3, 1.2, 3)
            if s % 5 == 0:
                half[s % 3] = 0.0                       # flat
            q = rng.normal(size=4) if s % 4 else np.array([1.0, 0, 0, 0])
            q /= np.linalg.norm(q)
            attrs = (*pos, *(pos - half), *(pos + half), *q)
            c.shapes[s][0] = 1
        else:
            attrs = (*pos, rng.uniform(0.3, 1.0))
            c.shapes[s][0] = 0
        c.shapes[s][1] = s
        c.shapes[s][2] = off
        for k, v in enumerate(attrs):
            c.attrs[off + k] = v
        off += len(attrs)
    assert off == 425
    c.num_shapes = 50
    c.num_imp = 10
    for k in range(10):
        c.imp[k] = 5 * k + 1 if k < 5 else 25 + 5 * (k - 5)   # five boxes, five spheres
    acc = O.new_accumulators(48, 32)
    n, glo, rays = O.render(c, acc, 2, 0, PIN_SEED, threads=4)
    ctx.set_constants(c)
    grays, gglo = ctx.render(2, ptk.SEED_REFERENCE, PIN_SEED)
    assert gglo == glo
    planes = ctx.read_planes()
    ok = _rel_ok(planes, acc, 1e-4).all(axis=0)
    assert ok.mean() >= 0.98
    # the oracle reports the last pass's bounce rays, the library the whole call's
    assert abs(grays - 2 * rays) <= 0.02 * 2 * rays
    with O.variant("R"):
        acc_r = O.new_accumulators(48, 32)
        O.render(c, acc_r, 2, 0, PIN_SEED, threads=4)
    assert _mismatch_report(planes, acc_r, 48) == ""


def random_pod_scene(O, seed, W=40, H=24, swap_bounds=False, degenerate=False):
    """A random scene built straight into the POD: 3 to 24 shapes (axis-aligned, flat and rotated
    boxes, several snapped to a coarse grid through the origin so that slab numerators and hit
    coordinates are exact zeros; spheres), all four material kinds, blur on some, lights as
    important shapes, camera inside the cloud of shapes, depth 1 to 12."""
    rng = np.random.default_rng(9000 + seed)
    depth = int(rng.integers(1, 13))
    sc = O.load_scene(scene_path("cornell"))
    c = O.build_constants(sc, W, H, depth, 0.005, True)
    n_mats = int(rng.integers(2, 12))
    for m in range(n_mats):
        kind = int(rng.integers(0, 4))
        col = rng.uniform(0.2, 1.0, 3) * (4.0 if kind == 1 else 1.0)
        blur = 0.05 if (kind >= 2 and rng.random() < 0.4) else 0.0
        for k, v in enumerate((*col, blur, float(rng.uniform(1.1, 1.8)), float(kind))):
            c.mats[m][k] = v
    c.mats[0][5] = 1.0                                    # at least one light
    n_shapes = int(rng.integers(3, 25))
    off = 0
    lights = []
    for s in range(n_shapes):
        mat = int(rng.integers(0, n_mats))
        snap = rng.random() < 0.4
        pos = rng.uniform(-5, 5, 3) + np.array([0.0, 0.0, 9.0])
        if snap:
            pos = np.round(pos)
        if rng.random() < 0.65:
            half = rng.uniform(0.3, 2.0, 3)
            if snap:
                half = np.maximum(np.round(half), 1.0)
            if rng.random() < 0.3:
                half[int(rng.integers(0, 3))] = 0.0       # flat
            q = rng.normal(size=4) if rng.random() < 0.4 else np.array([1.0, 0, 0, 0])
            q /= np.linalg.norm(q)
            lo, hi = pos - half, pos + half
            if swap_bounds and rng.random() < 0.5:        # a malformed box: lo > hi on one axis
                ax = int(rng.integers(0, 3))
                lo[ax], hi[ax] = hi[ax], lo[ax]
            attrs = (*pos, *lo, *hi, *q)
            c.shapes[s][0] = 1
        else:
            attrs = (*pos, float(rng.uniform(0.3, 1.5)))
            c.shapes[s][0] = 0
        c.shapes[s][1] = mat
        c.shapes[s][2] = off
        for k, v in enumerate(attrs):
            c.attrs[off + k] = v
        off += len(attrs)
        if int(c.mats[mat][5]) == 1:
            lights.append(s)
    assert off <= 450
    c.num_shapes = n_shapes
    lights = lights[:10]
    c.num_imp = len(lights)
    for k, s in enumerate(lights):
        c.imp[k] = s
    if degenerate:
        _make_degenerate(c, rng)
    return c


def _make_degenerate(c, rng):
    """Bends a POD scene towards exact zeros in the slab test: the jitter off (in an even frame
    the camera rays of the centre column and row then have a direction component of exactly +0),
    faces of unrotated boxes moved into the camera's planes (plane distances of 0 / 0, which the
    reference's comparisons turn into hits on boxes the ray never touches), and now and then the
    whole scene carried far from the origin, where `camera + offset - camera` rounds small
    offsets to zero by itself."""
    mode = int(rng.integers(0, 4))
    if mode != 3:
        c.rand_samp = 0.0
    if mode >= 2:
        shift = np.float32([rng.choice([-1, 1]) * 4000.0, 250.0, rng.choice([-1, 1]) * 3000.0])
        for k in range(3):
            c.cam_pos[k] = np.float32(c.cam_pos[k]) + shift[k]
        for s in range(c.num_shapes):
            off = c.shapes[s][2]
            for k in range(3):
                c.attrs[off + k] = np.float32(c.attrs[off + k]) + shift[k]
                if c.shapes[s][0] == 1:
                    c.attrs[off + 3 + k] = np.float32(c.attrs[off + 3 + k]) + shift[k]
                    c.attrs[off + 6 + k] = np.float32(c.attrs[off + 6 + k]) + shift[k]
    for s in range(c.num_shapes):
        off = c.shapes[s][2]
        if c.shapes[s][0] != 1 or c.attrs[off + 9] != 1.0 or rng.random() < 0.4:
            continue
        ax = int(rng.integers(0, 3))
        lo_face = rng.random() < 0.5
        c.attrs[off + (3 if lo_face else 6) + ax] = c.cam_pos[ax]
        if c.attrs[off + 3 + ax] > c.attrs[off + 6 + ax]:
            c.attrs[off + 3 + ax], c.attrs[off + 6 + ax] = c.attrs[off + 6 + ax], c.attrs[off + 3 + ax]


def random_scene_mismatch(ctx, O, seed, W=40, H=24, swap_bounds=False, degenerate=False):
    import ptk
    c = random_pod_scene(O, seed, W, H, swap_bounds, degenerate)
    with O.variant("R"):
        acc = O.new_accumulators(W, H)
        n, glo, rays = O.render(c, acc, 2, 0, PIN_SEED, threads=4)
    ctx.set_constants(c)
    grays, gglo = ctx.render(2, ptk.SEED_REFERENCE, PIN_SEED)
    if gglo != glo:
        return "generator state differs"
    return _mismatch_report(ctx.read_planes(), acc, W)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_random_scenes_are_bit_identical_to_the_correctly_rounded_restatement(ctx, oracle, seed):
    """Twelve random scenes (random_pod_scene): after two samples with the reference seeds every
    plane of every pixel equals variant R of the restatement bit for bit, and the generator
    state matches.  scripts/gpu_scene_fuzz.py runs the same check over hundreds of seeds."""
    assert random_scene_mismatch(ctx, oracle, seed) == ""


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(12))
def test_random_scenes_with_exact_zeros_in_the_slab_test(ctx, oracle, seed):
    """The same bar for scenes bent towards 0 / 0 plane distances (_make_degenerate): jitter off,
    box faces in the camera's planes, scenes 4000 units from the origin."""
    assert random_scene_mismatch(ctx, oracle, seed, degenerate=True) == ""


def test_reference_seeds_staged_in_row_bands_give_the_same_frame(oracle, monkeypatch):
    """A pass of reference seeds larger than the staging chunk is rendered in bands of rows
    (multiples of 16), the global recurrence running on across bands in pixel order: planes,
    seed state and ray count equal those of whole-pass staging, with and without features."""
    import ptk
    O = oracle
    W, H = 37, 50                                        # bands of 16, 16, 16 and 2 rows
    sc = O.load_scene(scene_path("MultiLightGlass"))
    results = []
    for chunk in (None, 37 * 16 * 16 + 100, 16):          # whole passes; 16-row bands; below one band
        if chunk is None:
            monkeypatch.delenv("PTK_SEED_CHUNK_BYTES", raising=False)
        else:
            monkeypatch.setenv("PTK_SEED_CHUNK_BYTES", str(chunk))
        c = ptk.Context(0)
        try:
            per_feat = []
            for feats in (False, True):
                c.set_constants(O.build_constants(sc, W, H, 4, 0.005, feats))
                rays, glo = c.render(3, ptk.SEED_REFERENCE, PIN_SEED)
                rays2, glo2 = c.render(2, ptk.SEED_REFERENCE, glo)      # a second call continues
                per_feat.append((rays + rays2, glo2, c.read_planes()))
            results.append(per_feat)
        finally:
            c.close()
    for other in results[1:]:
        for (r0, g0, p0), (r1, g1, p1) in zip(results[0], other):
            assert r0 == r1 and g0 == g1
            assert np.array_equal(p0.view(np.uint32), p1.view(np.uint32))
    acc = O.new_accumulators(W, H)
    with O.variant("R"):
        n, glo, _ = O.render(O.build_constants(sc, W, H, 4, 0.005, True), acc, 5, 0, PIN_SEED, threads=4)
    assert glo == results[0][1][1]
    assert _same(results[1][1][2], acc).all()


@pytest.mark.parametrize("name", ["MultiLightGlass", "cornell_crystal", "Box_Grid"])
def test_work_items_of_any_chunk_length_give_the_same_planes(oracle, monkeypatch, name):
    """A launch is cut into work items (a 16x16 tile x a chunk of samples) that resident
    workgroups take from a counter; a tile's chunks run in order, each waiting for the one
    before it.  On a small frame with chunks of 1, 3 and 8 samples nearly every item has to
    wait for its predecessor, which another workgroup is still tracing: planes (features
    included), ray count and, with reference seeds, the seed state must equal those of a
    launch with one item per tile, bit for bit -- for the fixed-size kernels with and without
    the blur lobe and for a hierarchy scene."""
    import ptk
    import scenes
    O = oracle
    W, H = 72, 40                                        # 5 x 3 tiles, the last row and column ragged
    sc = O.load_scene(scene_path(name))
    results = []
    for chunk in (None, 1, 3, 8):
        if chunk is None:
            monkeypatch.setenv("PTK_RENDER_CHUNK_SPP", "1000000")    # one item per tile
        else:
            monkeypatch.setenv("PTK_RENDER_CHUNK_SPP", str(chunk))
        c = ptk.Context(0)
        try:
            got = []
            for feats in (False, True):
                c.set_constants(O.build_constants(sc, W, H, 4, 0.005, feats))
                rays, glo = c.render(19, ptk.SEED_REFERENCE, PIN_SEED)
                rays2, _ = c.render(13, ptk.SEED_COUNTER)
                got.append((rays + rays2, glo, c.read_planes()))
            if name == "Box_Grid":                                   # and through a hierarchy
                c.set_scene(scenes.random_scene(300, W, H, seed=7), True)
                rays, glo = c.render(11, ptk.SEED_REFERENCE, PIN_SEED)
                got.append((rays, glo, c.read_planes()))
            results.append(got)
        finally:
            c.close()
    for other in results[1:]:
        for (r0, g0, p0), (r1, g1, p1) in zip(results[0], other):
            assert r0 == r1 and g0 == g1
            assert np.array_equal(p0.view(np.uint32), p1.view(np.uint32))


def test_work_item_hand_over_at_full_occupancy(oracle, monkeypatch):
    """The same at a size that fills the chip (1280x720: 3600 tiles on about 1500 resident
    workgroups, spread over all eight L2s), chunks of 8 samples: a tile's chunks are traced
    by whichever workgroup is free, so every hand-over of the sums crosses compute units and
    mostly L2s.  Colour and all feature planes bit-identical to one item per tile."""
    import ptk
    O = oracle
    W, H = 1280, 720
    sc = O.load_scene(scene_path("MultiLightGlass"))
    got = []
    for chunk in ("1000000", "8"):
        monkeypatch.setenv("PTK_RENDER_CHUNK_SPP", chunk)
        c = ptk.Context(0)
        try:
            c.set_constants(O.build_constants(sc, W, H, 4, 0.005, True))
            rays, _ = c.render(48, ptk.SEED_COUNTER)
            got.append((rays, c.read_planes()))
        finally:
            c.close()
    assert got[0][0] == got[1][0]
    assert np.array_equal(got[0][1].view(np.uint32), got[1][1].view(np.uint32))


@pytest.mark.parametrize("seed", range(50, 56))
def test_boxes_with_swapped_bounds_take_the_reference_comparison_chain(ctx, oracle, seed):
    """The slab test merges the three axes with max / min when every box of the scene has
    lo <= hi; a scene with a malformed box (lo > hi on an axis) must fall back to the
    reference's chain of comparisons, whose outcome for such a box is its own: every plane
    bit-identical to the correctly rounded restatement."""
    r = random_scene_mismatch(ctx, oracle, seed, swap_bounds=True)
    assert not r, r


@pytest.mark.parametrize("name", ["cornell", "Box_Grid", "basic"])
def test_both_kernel_families_render_a_scene_without_refraction_alike(oracle, monkeypatch, name):
    """Scenes in which no shape refracts are rendered by kernels whose shadow scans carry no
    bending code; PTK_SEGMENT_SCANS=1 keeps a context on the kernels for scenes that do refract
    (shadow scans by stretches between refracting shapes: here one stretch).  Same planes, ray
    count and seed state, features included."""
    import ptk
    O = oracle
    W, H = 96, 54
    consts = O.build_constants(O.load_scene(scene_path(name)), W, H, 4, 0.005, True)
    out = []
    for segments in (False, True):
        if segments:
            monkeypatch.setenv("PTK_SEGMENT_SCANS", "1")
        else:
            monkeypatch.delenv("PTK_SEGMENT_SCANS", raising=False)
        c = ptk.Context(0)
        try:
            c.set_constants(consts)
            rays, glo = c.render(3, ptk.SEED_REFERENCE, PIN_SEED)
            out.append((rays, glo, c.read_planes()))
        finally:
            c.close()
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1]
    assert np.array_equal(out[0][2].view(np.uint32), out[1][2].view(np.uint32))


def test_refracting_shapes_in_the_middle_of_the_index_order(ctx, oracle):
    """Shadow scans take the stretches between refracting shapes one after the other.  The scene
    files put their glass last; here glass boxes and spheres sit at the front, in the middle and
    next to each other, so that rays are bent before most shapes are tested: every plane
    bit-identical to the correctly rounded restatement."""
    import ptk
    O = oracle
    W, H = 48, 32
    for seed in range(4):
        c = random_pod_scene(O, 300 + seed, W, H)
        n = c.num_shapes
        glass = [m for m in range(50) if int(c.mats[m][5]) == 3]
        if not glass:
            c.mats[1][5] = 3.0
            c.mats[1][4] = 1.5
            glass = [1]
        for s in (0, 1, n // 2, n - 2):                   # front, neighbours, middle, near the end
            if 0 <= s < n:
                c.shapes[s][1] = glass[s % len(glass)]
        with O.variant("R"):
            acc = O.new_accumulators(W, H)
            _, glo, _ = O.render(c, acc, 2, 0, PIN_SEED, threads=4)
        ctx.set_constants(c)
        _, gglo = ctx.render(2, ptk.SEED_REFERENCE, PIN_SEED)
        assert gglo == glo
        assert _mismatch_report(ctx.read_planes(), acc, W) == ""
