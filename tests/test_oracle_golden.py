"""The oracle against the committed golden fixtures (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, scene_path, weights_path

SCENES = ["cornell", "cornell_crystal", "MultiLightGlass", "Box_Grid", "blur_test", "bars3"]


def test_rng_known_answers(oracle):
    g = np.load(os.path.join(GOLDEN, "rng_kat.npz"))
    seed = tuple(int(x) for x in g["seed"])
    seeds, state = oracle.seed_stream(seed, 64)
    assert np.array_equal(seeds, g["pixel_seeds"])
    assert state == tuple(int(x) for x in g["state_after"])
    d, _ = oracle.rand_stream(int(seeds[0, 0]), int(seeds[0, 1]), 64)
    assert np.array_equal(d.view(np.uint32), g["draws01_bits"])
    d, _ = oracle.rand_stream(int(seeds[1, 0]), int(seeds[1, 1]), 64, -1.0, 1.0)
    assert np.array_equal(d.view(np.uint32), g["draws_pm1_bits"])
    assert (d >= -1.0).all() and (d <= 1.0).all()


def test_rng_first_step_by_hand():
    """One step of each recurrence, worked in Python integers."""
    M = (1 << 64) - 1
    s0, s1 = 0x9E3779B97F4A7C15, 0xD1B54A32D192ED03
    b = s1 ^ s0
    g0 = ((s0 << 49) & M) | ((s0 >> 15) ^ b ^ ((b << 21) & M))
    g1 = ((b << 28) & M) | (b >> 36)
    g = np.load(os.path.join(GOLDEN, "rng_kat.npz"))
    assert int(g["pixel_seeds"][0, 0]) & M == g0
    assert int(g["pixel_seeds"][0, 1]) & M == g1
    # randBetween(0,1) from (g0,g1): rotl(g0+g1,17)+g0, /2^64, to float
    t = (g0 + g1) & M
    word = ((((t << 17) & M) | (t >> 47)) + g0) & M
    expect = np.float32(np.float64(word) / 2.0 ** 64)
    assert g["draws01_bits"][0] == expect.view(np.uint32)


@pytest.mark.parametrize("name", SCENES)
def test_render_fixtures(oracle, name):
    O = oracle
    g = np.load(os.path.join(GOLDEN, "render_%s_64x36.npz" % name))
    W, H = 64, 36
    sc = O.load_scene(scene_path(name))
    consts = O.build_constants(sc, W, H, 4, 0.005, True)
    assert bytes(consts) == g["constants"].tobytes()
    acc = O.new_accumulators(W, H)
    seed = (0x9E3779B97F4A7C15, 0xD1B54A32D192ED03)
    n, glo, rays = O.render(consts, acc, 8, 0, seed, threads=4)
    assert n == int(g["sample_count"]) and rays == int(g["rays_last"])
    assert glo == tuple(int(x) for x in g["glo_after"])
    assert np.array_equal(acc, g["acc_8spp"], equal_nan=True)
    for f in (0, 1, 2):
        assert O.auto_exposure(acc, W, H, n, f)[0] == g["exposure"][f]
    # spot-check single samples
    pix, _ = O.seed_stream(seed, W * H)
    for idx in (0, 1, 63, 64, 1000, W * H - 1):
        s = O.render_sample(consts, idx // W, idx % W, int(pix[idx, 0]), int(pix[idx, 1]))
        got = np.array(list(s.xyz) + list(s.normal) + list(s.albedo1) + list(s.albedo2)
                       + list(s.world_pos) + [s.direct_light, s.rays_sent], dtype=np.float32)
        assert np.array_equal(got, g["samples_1spp"][idx], equal_nan=True)


def test_denoise_fixtures(oracle):
    O = oracle
    g = np.load(os.path.join(GOLDEN, "denoise_MultiLightGlass_128x72.npz"))
    W, H = 128, 72
    acc = np.ascontiguousarray(g["acc_8spp"])
    n = int(g["sample_count"])
    w = O.load_weights(weights_path("BEST"))
    feats, var = O.forward_prop(acc, W, H, n, w)
    assert np.array_equal(feats, g["features36"])
    assert np.array_equal(var, g["variances7"])
    for nwin in (1, 3):
        d, wc = O.filter(acc, var, W, H, n, nwin, threads=4)
        assert np.array_equal(d, g["denoised_n%d" % nwin])
        assert np.array_equal(wc, g["wcsum_n%d" % nwin])


def test_weight_files_have_530_values():
    for name in ("BEST", "init", "box_grid", "NNWeights"):
        lines = open(weights_path(name)).read().splitlines()
        assert [len(l.split()) for l in lines[:3]] == [360, 100, 70]


def test_post_process_metric0_is_the_tonemap(oracle):
    """The general display transform at metric 0, quantised, equals the mode-0 tone-map."""
    g = np.load(os.path.join(GOLDEN, "render_cornell_64x36.npz"))
    acc = g["acc_8spp"]
    post = oracle.post_process(acc, None, 64, 36, 8, 0, 0.8, 2.2)
    assert np.array_equal(oracle.quantise8(post, 64, 36).reshape(-1),
                          np.asarray(oracle.tonemap8(acc, 64, 36, 8, 0.8, 2.2)).reshape(-1))
    with pytest.raises(ValueError):
        oracle.post_process(acc, None, 64, 36, 8, 1, 1.0, 2.2)    # metric 1 needs a denoised image
    with pytest.raises(ValueError):
        oracle.post_process(acc, None, 64, 36, 8, 7, 1.0, 2.2)    # so does metric 7 (the target)
    with pytest.raises(ValueError):
        oracle.post_process(acc, None, 64, 36, 8, 8, 1.0, 2.2)
    # metrics 1 and 7 display the given screen as it is
    screen = np.ascontiguousarray(acc[0:3].T / 8)
    assert np.array_equal(oracle.post_process(acc, screen, 64, 36, 8, 7, 0.8, 2.2), post)
    # metric 2 maps normals from [-1, 1] to [0, 1]
    n = oracle.post_process(acc, None, 64, 36, 8, 2, 1.0, 1.0)
    assert np.nanmin(n) >= -1e-6 and np.nanmax(n) <= 1.0 + 1e-6


def test_training_restatement_structure(oracle):
    """GenRelMSE / CPUBackProp restatement ("parity unpinned": the reference recorded no
    training output): relMSE vanishes on a perfect target, the update is linear in the learning
    rate, repeats with each layer's fan-in, and portrait images (where the reference's filter
    derivative indexes outside the image) are refused."""
    O = oracle
    W, H, spp = 40, 24, 6
    sc = O.load_scene(scene_path("cornell"))
    acc = O.new_accumulators(W, H)
    O.render(O.build_constants(sc, W, H, 4, 0.005, True), acc, spp, 0, (11, 12), threads=4)
    w = O.load_weights(weights_path("box_grid"))
    feats, var, l2, l3 = O.forward_prop_layers(acc, W, H, spp, w, threads=2)
    f0, v0 = O.forward_prop(acc, W, H, spp, w, threads=2)
    assert np.array_equal(feats, f0) and np.array_equal(var, v0)
    assert np.all((l2 >= 0) & (l2 <= 1)) and np.all((l3 >= 0) & (l3 <= 1))   # sigmoid outputs
    den, wc = O.filter(acc, var, W, H, spp, 1)
    assert O.rel_mse(den, den, W, H, spp) == 0.0
    target = (acc[0:3].T / spp) * 0.8 + 0.05
    assert O.rel_mse(den, target, W, H, spp) > 0.0
    zero = tuple(np.zeros_like(x) for x in w)
    d1 = O.back_prop(acc, den, wc, var, feats, l2, l3, target, W, H, spp, 1, 1e-3, zero)
    d2 = O.back_prop(acc, den, wc, var, feats, l2, l3, target, W, H, spp, 1, 2e-3, zero)
    for a, b, fan in zip(d1, d2, (36, 10, 10)):
        assert np.isfinite(a).all() and np.abs(a).max() > 0
        assert np.allclose(2 * a, b, rtol=1e-4, atol=1e-9)
        assert np.allclose(a[:fan], a[fan:2 * fan], rtol=1e-5, atol=1e-9)
    same = O.back_prop(acc, den, wc, var, feats, l2, l3, target, W, H, spp, 1, 0.0, w)
    assert all(np.array_equal(a, b) for a, b in zip(same, w))
    with pytest.raises(ValueError):
        O.back_prop(acc, den, wc, var, feats, l2, l3, target, H, W, spp, 1, 1e-3, w)
