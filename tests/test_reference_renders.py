"""The CPU restatement against an image the reference's authors rendered themselves: the one
check in this repository pinned to a reference-held OUTPUT at pixel granularity (see
tests/reference_renders.py for what is fixed, what is fitted and why it is statistical)."""
import os

import numpy as np
import pytest

import reference_renders as RR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as O
    O.build()
    return O


def test_oracle_reproduces_the_reference_128_spp_render_within_its_own_noise(oracle):
    O = oracle
    spp = 128
    ref = RR.load(spp)
    sc = O.load_scene(os.path.join(ROOT, "data", "scenes", "MultiLightGlass.scene"))
    consts = O.build_constants(sc, RR.WIDTH, RR.HEIGHT, RR.FILES[spp], 0.005, False)
    imgs = []
    for seed in ((0x9E3779B97F4A7C15, 0xD1B54A32D192ED03), (0x0123456789ABCDEF, 0x0F1E2D3C4B5A6978)):
        acc = O.new_accumulators(RR.WIDTH, RR.HEIGHT)
        n, _, _ = O.render(consts, acc, spp, 0, seed, threads=min(8, os.cpu_count() or 1))
        assert n == spp
        imgs.append(RR.file_order(acc[0:3], spp))
    RR.check(imgs[0], imgs[1], ref, "oracle, %d spp, depth %d" % (spp, RR.FILES[spp]))


def test_the_comparison_rejects_a_wrong_path_depth(oracle):
    """The statistic has teeth: the same scene traced four bounces deep instead of twelve is
    far outside the tolerance (16 x 16 block means differ by 20 levels, not 1)."""
    O = oracle
    spp = 16
    ref = RR.load(128)
    sc = O.load_scene(os.path.join(ROOT, "data", "scenes", "MultiLightGlass.scene"))
    consts = O.build_constants(sc, RR.WIDTH, RR.HEIGHT, 4, 0.005, False)
    acc = O.new_accumulators(RR.WIDTH, RR.HEIGHT)
    O.render(consts, acc, spp, 0, (1, 2), threads=min(8, os.cpu_count() or 1))
    img = RR.file_order(acc[0:3], spp)
    e = RR.fit_exposure(img, ref)
    assert not (0.218 < e < 0.228)
    assert RR.block_diff(RR.quantise(img, e), ref) > 8.0


def test_oracle_reproduces_the_reference_edgeblur_render_within_its_own_noise(oracle):
    """Renders/EdgeBlur.ppm: the cornell scene (boxes, a rotated box, a box light), 384x216."""
    O = oracle
    E = RR.EDGEBLUR
    sc = O.load_scene(os.path.join(ROOT, "data", "scenes", E["scene"] + ".scene"))
    consts = O.build_constants(sc, E["width"], E["height"], E["depth"], 0.005, False)
    imgs = []
    for seed in ((0x9E3779B97F4A7C15, 0xD1B54A32D192ED03), (0x0123456789ABCDEF, 0x0F1E2D3C4B5A6978)):
        acc = O.new_accumulators(E["width"], E["height"])
        n, _, _ = O.render(consts, acc, E["spp"], 0, seed, threads=min(8, os.cpu_count() or 1))
        imgs.append(RR.edgeblur_file_order(acc[0:3], n))
    RR.check_edgeblur(imgs[0], imgs[1], "oracle, cornell %d spp, depth %d" % (E["spp"], E["depth"]))
