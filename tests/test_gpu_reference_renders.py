"""The GPU path against the images the reference's authors rendered themselves (5000, 2500 and
128 samples per pixel): each file must be as close to a render of ours as a second,
independently seeded render of ours is, per pixel and in 16 x 16 block means, with one fitted
scalar (the unrecorded exposure).  See tests/reference_renders.py."""
import os
import sys

import numpy as np
import pytest

import reference_renders as RR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import ptk
    c = ptk.Context(0)
    yield c
    c.close()


def _render(ctx, spp, depth, stream):
    import ptk
    raw, _, _ = ptk.scene_constants(os.path.join(ROOT, "data", "scenes", "MultiLightGlass.scene"),
                                    RR.WIDTH, RR.HEIGHT, depth, 0.005, False)
    ctx.set_constants(raw)
    ctx.set_stream_id(stream)
    ctx.render(spp, ptk.SEED_COUNTER)
    return RR.file_order(ctx.read_planes(0, 3), spp)


@pytest.mark.parametrize("spp", sorted(RR.FILES))
def test_reference_render_is_within_our_own_noise(ctx, spp):
    depth = RR.FILES[spp]
    first, second = _render(ctx, spp, depth, 0), _render(ctx, spp, depth, 1)
    e, b_ref, b_self = RR.check(first, second, RR.load(spp), "GPU, %d spp, depth %d" % (spp, depth))
    if spp == 5000:
        # teeth: eight bounces instead of eleven is five times our block noise away
        shallow = _render(ctx, spp, 8, 0)
        q = RR.quantise(shallow, RR.fit_exposure(shallow, RR.load(spp)))
        assert RR.block_diff(q, RR.load(spp)) > 3.0 * b_self
    ctx.set_stream_id(0)


def test_reference_edgeblur_render_is_within_our_own_noise(ctx):
    """Renders/EdgeBlur.ppm against the cornell scene: per pixel, in block means and in its
    amount of pixel noise; four bounces instead of twelve are far outside."""
    import ptk
    E = RR.EDGEBLUR

    def render(depth, stream):
        raw, _, _ = ptk.scene_constants(os.path.join(ROOT, "data", "scenes", E["scene"] + ".scene"),
                                        E["width"], E["height"], depth, 0.005, False)
        ctx.set_constants(raw)
        ctx.set_stream_id(stream)
        ctx.render(E["spp"], ptk.SEED_COUNTER)
        return RR.edgeblur_file_order(ctx.read_planes(0, 3), E["spp"])

    first, second = render(E["depth"], 0), render(E["depth"], 1)
    _, _, b_self = RR.check_edgeblur(first, second, "GPU, cornell %d spp, depth %d" % (E["spp"], E["depth"]))
    shallow = render(4, 0)
    ref = RR.load_edgeblur()
    q = RR.quantise(shallow, RR.fit_exposure_blocks(shallow, ref))
    assert RR.block_diff_any(q, ref) > 2.0 * b_self
    ctx.set_stream_id(0)
