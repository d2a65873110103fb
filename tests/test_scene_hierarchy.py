"""Host side of the scalable scene path (no GPU): scene validation, the hierarchy
builder's invariants (through ptk_check_scene, which verifies them inside the
library), and the CPU checker's any-size scan against its fixed-size one."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, PIN_SEED


@pytest.mark.parametrize("n", [0, 1, 2, 3, 7, 64, 1000, 30000])
def test_hierarchy_invariants_random_scenes(n):
    import ptk
    import scenes
    st = ptk.check_scene(scenes.random_scene(n, 64, 36, seed=n + 1))
    assert st["num_shapes"] == n and st["leaves"] == n
    assert st["nodes"] == max(1, n - 1)          # one shape per leaf, two children per node
    assert st["depth"] <= 62
    if n >= 2:
        assert st["depth"] >= int(np.ceil(np.log2(n)))


def test_hierarchy_invariants_degenerate_layouts():
    import ptk
    import scenes
    base = scenes.random_scene(400, 64, 36, seed=9)
    # all shapes on one spot: no split plane separates them, the builder falls back to medians
    same = dict(base)
    same["shape_attrs"] = np.repeat(base["shape_attrs"][5:6], 400, axis=0)
    same["shape_type"] = np.repeat(base["shape_type"][5:6], 400)
    st = ptk.check_scene(same)
    assert st["leaves"] == 400 and st["depth"] <= 62
    # a line of shapes with geometrically growing gaps: what unbalances a surface-area split
    line = dict(base)
    sa = np.zeros((400, 13), np.float32)
    sa[:, 0] = 1.02 ** np.arange(400)
    sa[:, 3] = 0.001
    line["shape_attrs"] = sa
    line["shape_type"] = np.zeros(400, np.int32)
    st = ptk.check_scene(line)
    assert st["leaves"] == 400 and st["depth"] <= 62


def test_bad_scenes_are_rejected_on_the_host():
    import ptk
    import scenes
    sc = scenes.random_scene(40, 32, 16, seed=5)
    ptk.check_scene(sc)

    def broken(**kw):
        bad = dict(sc)
        for k, v in kw.items():
            bad[k] = v
        return bad

    mats = sc["shape_mat"].copy()
    mats[3] = 99
    with pytest.raises(ptk.PtkError, match="material"):
        ptk.check_scene(broken(shape_mat=mats))
    with pytest.raises(ptk.PtkError, match="important"):
        ptk.check_scene(broken(important_shapes=np.asarray([40], np.uint32)))
    box = int(np.nonzero(sc["shape_type"] == 1)[0][0])
    attrs = sc["shape_attrs"].copy()
    attrs[box, 9:13] = [2.0, 0.0, 0.0, 0.0]
    with pytest.raises(ptk.PtkError, match="quaternion"):
        ptk.check_scene(broken(shape_attrs=attrs))
    attrs = sc["shape_attrs"].copy()
    attrs[box, 0] = np.nan
    with pytest.raises(ptk.PtkError, match="finite"):
        ptk.check_scene(broken(shape_attrs=attrs))
    attrs = sc["shape_attrs"].copy()
    attrs[box, 0] = 3e7
    with pytest.raises(ptk.PtkError, match="exceed"):
        ptk.check_scene(broken(shape_attrs=attrs))
    types = sc["shape_type"].copy()
    types[0] = 2
    with pytest.raises(ptk.PtkError, match="type"):
        ptk.check_scene(broken(shape_type=types))
    with pytest.raises(ptk.PtkError, match="max_depth"):
        ptk.check_scene(broken(max_depth=13))
    with pytest.raises(ptk.PtkError, match="resolution"):
        ptk.check_scene(broken(res_h=0))
    with pytest.raises(ptk.PtkError, match="flags"):
        ptk.check_scene(sc, flags=8)


@pytest.mark.parametrize("name", ["cornell_crystal", "MultiLightGlass", "blur_test"])
def test_any_size_scan_equals_fixed_size_scan_on_cpu(oracle, name):
    """orc_render_scene (tables of any length) against orc_render (the POD) on the
    reference's own scenes: the same per-sample function, so the same bits."""
    import ptk
    g = np.load(os.path.join(GOLDEN, "render_%s_64x36.npz" % name))
    raw = g["constants"].tobytes()
    consts = oracle.Constants.from_buffer_copy(raw)
    consts.want_features = 1
    a = oracle.new_accumulators(64, 36)
    sc_a, glo_a, rays_a = oracle.render(consts, a, 2, 0, PIN_SEED, threads=4)
    b = oracle.new_accumulators(64, 36)
    sc_b, glo_b, rays_b = oracle.render_scene(ptk.scene_from_constants(raw), b, 2, 0, PIN_SEED,
                                              want_features=True, threads=4)
    assert (sc_a, glo_a, rays_a) == (sc_b, glo_b, rays_b)
    assert ((a == b) | (np.isnan(a) & np.isnan(b))).all()
