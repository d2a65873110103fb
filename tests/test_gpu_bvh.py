"""GPU parity of the scalable scene path (ptk_set_scene: shapes in HBM behind a
bounding-volume hierarchy) through the C ABI.

What is checked, and to what bar:

  * The reference's own scenes pushed through ptk_set_scene give the SAME BITS in
    every plane as through ptk_set_constants (the fixed-size path whose parity
    with the CPU checker test_gpu_parity.py establishes): the hierarchy changes
    which shapes a ray is tested against, never a result.
  * Scenes of hundreds to thousands of shapes give the same bits with the
    hierarchy as with PTK_SCENE_VISIT_ALL (every ray tests every shape), with
    and without refracting shapes and blur: nothing the index-order scan would
    accept is culled.
  * Against the CPU checker's index-order scan over the same tables
    (orc_render_scene: the reference's per-sample function with the 50-shape
    limit lifted -- "parity unpinned" beyond 50 shapes, since the reference
    cannot hold such a scene): >= 99 % of pixels bit-identical per sample to the
    canonical build (glibc's faithfully rounded cosf/sinf/acosf/powf), and EVERY
    plane of EVERY pixel bit-identical to the build with those four calls
    correctly rounded (variant R) -- the same two bars as test_gpu_parity.py.
"""
import numpy as np
import pytest

from conftest import GOLDEN, PIN_SEED

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import ptk
    c = ptk.Context(0)
    yield c
    c.close()


def _same(a, b):
    return (a == b) | (np.isnan(a) & np.isnan(b))


@pytest.mark.parametrize("name", ["cornell", "cornell_crystal", "MultiLightGlass", "Box_Grid",
                                  "blur_test", "bars3"])
@pytest.mark.parametrize("math_mode", [0, 1])
def test_reference_scenes_same_bits_as_fixed_size_path(ctx, name, math_mode):
    import os
    import ptk
    g = np.load(os.path.join(GOLDEN, "render_%s_64x36.npz" % name))
    raw = g["constants"].tobytes()
    ctx.set_math_mode(math_mode)
    ctx.set_constants(raw, want_features=1)
    rays_a, glo_a = ctx.render(5, ptk.SEED_REFERENCE, PIN_SEED)
    a = ctx.read_planes()
    stats = ctx.set_scene(ptk.scene_from_constants(raw), want_features=1)
    assert stats["num_shapes"] == np.frombuffer(raw, np.int32)[14]
    rays_b, glo_b = ctx.render(5, ptk.SEED_REFERENCE, PIN_SEED)
    b = ctx.read_planes()
    ctx.set_math_mode(ptk.MATH_EXACT)
    assert rays_a == rays_b and glo_a == glo_b
    assert _same(a, b).all()


@pytest.mark.parametrize("n,dielectric,blur,depth", [(300, True, False, 4), (300, True, True, 4),
                                                     (2000, False, False, 4), (2000, True, False, 8),
                                                     (5000, True, True, 12)])
def test_hierarchy_culls_nothing_the_scan_would_accept(ctx, n, dielectric, blur, depth):
    import ptk
    import scenes
    sc = scenes.random_scene(n, 96, 64, seed=n + depth, dielectric=dielectric, blur=blur,
                             max_depth=depth)
    stats = ctx.set_scene(sc, want_features=1)
    assert stats["num_shapes"] == n and stats["depth"] <= 62 and stats["leaves"] >= n // 4
    rays_a, _ = ctx.render(3, ptk.SEED_REFERENCE, PIN_SEED)
    a = ctx.read_planes()
    ctx.set_scene(sc, want_features=1, flags=ptk.SCENE_VISIT_ALL)
    rays_b, _ = ctx.render(3, ptk.SEED_REFERENCE, PIN_SEED)
    b = ctx.read_planes()
    assert rays_a == rays_b
    assert _same(a, b).all()
    assert np.isfinite(a[0:3]).all() and a[0:3].mean() > 0


def _face_plane_scene(seed, shift):
    """400 shapes, no jitter, an even frame: the camera rays of the centre column have a
    direction whose x is exactly +0 and those of the centre row one whose y is.  Sixty small
    boxes get a face exactly in the camera's x or y plane (shifted by `shift`), away from
    where those rays go."""
    import scenes
    W, H = 64, 48
    sc = scenes.random_scene(400, W, H, seed=seed, dielectric=True)
    sc["rand_samp"] = 0.0
    rng = np.random.default_rng(seed + 1)
    st, sa = sc["shape_type"], sc["shape_attrs"]
    cam = sc["cam_pos"]
    picks = rng.choice(np.arange(2, 400), 60, replace=False)
    for k, s in enumerate(picks):
        axis = k % 2                        # x plane / y plane
        upper = (k // 2) % 2                # the box's upper or lower face lies in it
        c = rng.uniform([-6, -3, 3], [6, 8, 25]).astype(np.float32)
        h = rng.uniform(0.1, 0.5, 3).astype(np.float32)
        lo, hi = c - h, c + h
        plane = np.float32(cam[axis] + shift)
        if upper:
            lo[axis], hi[axis] = plane - np.float32(0.4), plane
        else:
            lo[axis], hi[axis] = plane, plane + np.float32(0.4)
        st[s] = 1
        sa[s, :] = 0
        sa[s, 0:3] = (lo + hi) / 2
        sa[s, 3:6], sa[s, 6:9] = lo, hi
        sa[s, 9] = 1.0
        if k % 5 == 4:                      # rotated about that axis: the axis is kept
            a = rng.uniform(0.2, 1.2)
            sa[s, 9] = np.cos(a / 2)
            sa[s, 10 + axis] = np.sin(a / 2)
            sa[s, 0:3] = 0                  # ... about the origin's axis, so the plane stays put
            sa[s, axis] = plane
    return sc, W, H


def test_boxes_reached_only_through_a_zero_over_zero_plane_distance(ctx, oracle):
    """The reference's slab test divides by the direction (Renderers.cpp:157-196).  A ray whose
    direction has an exactly zero component, starting exactly in the plane of a box face on that
    axis, gets 0 / 0 for that plane; every comparison with the NaN fails and the test accepts the
    box whenever the other two axes' entry and exit distances are in ascending order -- wherever
    the box is.  An index-order scan returns such hits; a hierarchy that culls by bounds alone
    never reaches those boxes.  The walk visits them for exactly such rays: same bits as
    visit-all and as the CPU scan -- and the scene really has such hits, since moving the face
    planes by 0.001 changes the centre column and row."""
    import ptk
    out = {}
    for shift in (0.0, 1e-3):
        sc, W, H = _face_plane_scene(31, shift)
        for flags in (0, ptk.SCENE_VISIT_ALL):
            ctx.set_scene(sc, want_features=1, flags=flags)
            rays, _ = ctx.render(2, ptk.SEED_REFERENCE, PIN_SEED)
            out[shift, flags] = (ctx.read_planes(), rays)
        a, b = out[shift, 0], out[shift, ptk.SCENE_VISIT_ALL]
        assert a[1] == b[1] and _same(a[0], b[0]).all()
    sc, W, H = _face_plane_scene(31, 0.0)
    with oracle.variant("R"):
        acc = oracle.new_accumulators(W, H)
        oracle.render_scene(sc, acc, 2, 0, PIN_SEED, want_features=True, threads=8)
    assert _same(out[0.0, 0][0], acc).all()
    # the quirk is live: first-hit positions of the centre column / row depend on the exact planes
    pos0 = out[0.0, 0][0][12:15].reshape(3, H, W)
    pos1 = out[1e-3, 0][0][12:15].reshape(3, H, W)
    moved = (np.abs(pos0 - pos1) > 0.05).any(axis=0)
    assert moved[:, W // 2].sum() + moved[H // 2, :].sum() >= 4
    off_axis = moved.copy()
    off_axis[:, W // 2] = False
    off_axis[H // 2, :] = False
    assert off_axis.sum() <= moved[:, W // 2].sum() + moved[H // 2, :].sum()


@pytest.mark.parametrize("n,dielectric", [(120, True), (1500, True), (1500, False)])
def test_against_index_order_scan_on_cpu(ctx, oracle, n, dielectric):
    import ptk
    import scenes
    W, H = 64, 36
    sc = scenes.random_scene(n, W, H, seed=7 * n, dielectric=dielectric)
    ctx.set_scene(sc, want_features=1)
    rays, glo = ctx.render(1, ptk.SEED_REFERENCE, PIN_SEED)
    gpu = ctx.read_planes()
    acc = oracle.new_accumulators(W, H)
    _, glo_o, rays_o = oracle.render_scene(sc, acc, 1, 0, PIN_SEED, want_features=True, threads=8)
    assert glo == glo_o and rays == rays_o
    same = _same(gpu[0:3], acc[0:3]).all(axis=0)
    assert same.mean() >= 0.99
    assert _same(gpu[3:16], acc[3:16]).all(axis=0).mean() >= 0.999
    # four more samples: the accumulators agree where every sample agreed
    ctx.render(4, ptk.SEED_REFERENCE, glo)
    gpu = ctx.read_planes()
    oracle.render_scene(sc, acc, 4, 1, glo_o, want_features=True, threads=8)
    close = np.abs(gpu[0:3].astype(np.float64) - acc[0:3]) <= 1e-4 * np.abs(acc[0:3]) + 1e-7
    assert close.all(axis=0).mean() >= 0.97
    assert abs(gpu[0:3].mean() - acc[0:3].mean()) <= 2e-3 * acc[0:3].mean()
    with oracle.variant("R"):
        acc_r = oracle.new_accumulators(W, H)
        oracle.render_scene(sc, acc_r, 5, 0, PIN_SEED, want_features=True, threads=8)
    assert _same(gpu, acc_r).all()


def test_full_size_frame_spot_checked_against_the_cpu(ctx, oracle):
    """1920x1080, 20 000 shapes with glass, the reference's sequential seed stream: one pass on
    the GPU through the hierarchy; 3000 pixels drawn at random are recomputed on the CPU by the
    index-order scan over all 20 000 shapes from the same seeds.  Same bar as the small frames."""
    import ptk
    import scenes
    O = oracle
    W, H = 1920, 1080
    P = W * H
    sc = scenes.random_scene(20000, W, H, seed=77)
    stats = ctx.set_scene(sc, want_features=1)
    assert stats["leaves"] == 20000
    rays, glo = ctx.render(1, ptk.SEED_REFERENCE, PIN_SEED)
    gpu = ctx.read_planes(0, 16)
    seeds, glo_o = O.seed_stream(PIN_SEED, P)
    assert glo == glo_o
    picks = np.sort(np.random.default_rng(3).choice(P, 3000, replace=False))
    ref = O.render_scene_pixels(sc, picks // W, picks % W, seeds[picks], want_features=True,
                                threads=8)
    got = gpu[:, picks].T
    same = ((got == ref[:, :16]) | (np.isnan(got) & np.isnan(ref[:, :16]))).all(axis=1)
    assert same.mean() >= 0.99
    first_hit = (got[:, 3:15] == ref[:, 3:15]).all(axis=1)
    assert first_hit.mean() >= 0.999
    with O.variant("R"):
        ref_r = O.render_scene_pixels(sc, picks // W, picks % W, seeds[picks], want_features=True,
                                      threads=8)
    assert _same(got, ref_r[:, :16]).all()


def test_tables_beyond_every_pod_limit(ctx, oracle):
    """More than 50 shapes, more than 127 materials (the fixed-size kernel's record
    holds 7 bits), more than 10 important shapes, at once."""
    import ptk
    import scenes
    W, H = 64, 36
    sc = scenes.random_scene(900, W, H, seed=21, lights=40, extra_mats=400)
    assert len(sc["mat_list"]) > 400 and len(sc["important_shapes"]) == 40
    assert sc["shape_mat"].max() > 127
    ctx.set_scene(sc, want_features=1)
    rays, glo = ctx.render(2, ptk.SEED_REFERENCE, PIN_SEED)
    gpu = ctx.read_planes()
    ctx.set_scene(sc, want_features=1, flags=ptk.SCENE_VISIT_ALL)
    ctx.render(2, ptk.SEED_REFERENCE, PIN_SEED)
    assert _same(gpu, ctx.read_planes()).all()
    acc = oracle.new_accumulators(W, H)
    _, glo_o, rays_o = oracle.render_scene(sc, acc, 2, 0, PIN_SEED, want_features=True, threads=8)
    assert glo == glo_o
    assert _same(gpu[0:3], acc[0:3]).all(axis=0).mean() >= 0.98
    assert _same(gpu[6:9], acc[6:9]).all(axis=0).mean() >= 0.995    # first-hit albedo: the material
    with oracle.variant("R"):
        acc_r = oracle.new_accumulators(W, H)
        oracle.render_scene(sc, acc_r, 2, 0, PIN_SEED, want_features=True, threads=8)
    assert _same(gpu, acc_r).all()


@pytest.mark.parametrize("n", [0, 1, 3, 5])
def test_tiny_scenes(ctx, oracle, n):
    import ptk
    import scenes
    W, H = 32, 16
    sc = scenes.random_scene(n, W, H, seed=11, lights=1)
    stats = ctx.set_scene(sc)
    assert stats["num_shapes"] == n
    rays, _ = ctx.render(2, ptk.SEED_REFERENCE, PIN_SEED)
    gpu = ctx.read_planes(0, 3)
    acc = oracle.new_accumulators(W, H)
    oracle.render_scene(sc, acc, 2, 0, PIN_SEED, threads=1)
    assert _same(gpu, acc[0:3]).all(axis=0).mean() >= 0.99
    with oracle.variant("R"):
        acc_r = oracle.new_accumulators(W, H)
        oracle.render_scene(sc, acc_r, 2, 0, PIN_SEED, threads=1)
    assert _same(gpu, acc_r[0:3]).all()
    if n == 0:
        assert rays == 2 * W * H     # every path is one ray into the sky


def test_counter_seeds_and_statistics(ctx):
    import ptk
    import scenes
    sc = scenes.grid_scene(12, 128, 72)
    stats = ctx.set_scene(sc, want_features=1)
    assert stats["nodes"] > 100 and stats["device_bytes"] > 0
    ctx.render(4, ptk.SEED_COUNTER)
    a = ctx.read_planes()
    ctx.reset()
    ctx.render(1, ptk.SEED_COUNTER)
    ctx.render(3, ptk.SEED_COUNTER)
    b = ctx.read_planes()
    assert _same(a, b).all()      # split launches carry the running statistics
    assert ctx.sample_count == 4
    # the fixed-size path takes over again
    import os
    raw = np.load(os.path.join(GOLDEN, "render_cornell_64x36.npz"))["constants"].tobytes()
    ctx.set_constants(raw)
    ctx.render(1, ptk.SEED_REFERENCE, PIN_SEED)
    assert np.isfinite(ctx.read_planes(0, 3)).all()


def test_camera_moves_keep_the_scene(ctx):
    """ptk_set_camera: the same frame as setting the whole scene again with that camera, for
    both kinds of scene; a position outside the box the hierarchy was built for is refused."""
    import os
    import ptk
    import scenes
    sc = scenes.random_scene(700, 64, 36, seed=31)
    pos = np.float32([1.5, 2.0, -0.5])
    c, s_ = np.float32(np.cos(0.3)), np.float32(np.sin(0.3))
    fwd, right, up = np.float32([s_, 0, c]), np.float32([c, 0, -s_]), np.float32([0, 1, 0])
    ctx.set_scene(sc, want_features=1)
    ctx.render(1, ptk.SEED_REFERENCE, PIN_SEED)          # something to be reset
    ctx.set_camera(pos, fwd, right, up)
    assert ctx.sample_count == 0
    ctx.render(2, ptk.SEED_REFERENCE, PIN_SEED)
    a = ctx.read_planes()
    moved = dict(sc, cam_pos=pos, cam_forward=fwd, cam_right=right, cam_up=up)
    ctx.set_scene(moved, want_features=1)
    ctx.render(2, ptk.SEED_REFERENCE, PIN_SEED)
    assert _same(a, ctx.read_planes()).all()
    # the hierarchy holds for cameras within the bounds of the shapes and of the camera the scene
    # was set with (the spheres' bounds are padded for the longest ray that fits): beyond, refused
    with pytest.raises(ptk.PtkError, match="outside"):
        ctx.set_camera(np.float32([1e4, 0, 0]), fwd, right, up)
    top = float(moved["shape_attrs"][:, 1].max())
    with pytest.raises(ptk.PtkError, match="outside"):
        ctx.set_camera(np.float32([0, top + 40.0, 5.0]), fwd, right, up)
    ctx.set_camera(np.float32([0, 0.5 * top, 5.0]), fwd, right, up)      # inside the cloud of shapes
    ctx.render(1, ptk.SEED_REFERENCE, PIN_SEED)
    inside = ctx.read_planes()
    ctx.set_scene(dict(moved, cam_pos=np.float32([0, 0.5 * top, 5.0])), want_features=1)
    ctx.render(1, ptk.SEED_REFERENCE, PIN_SEED)
    assert _same(inside, ctx.read_planes()).all()

    raw = bytearray(np.load(os.path.join(GOLDEN, "render_cornell_crystal_64x36.npz"))["constants"].tobytes())
    ctx.set_constants(bytes(raw), want_features=1)
    ctx.set_camera(np.float32([0.5, -2.0, -8.0]), fwd, right, up)
    ctx.render(2, ptk.SEED_REFERENCE, PIN_SEED)
    a = ctx.read_planes()
    np.frombuffer(raw, np.float32)[0:12] = np.concatenate([np.float32([0.5, -2.0, -8.0]), fwd, right, up])
    ctx.set_constants(bytes(raw), want_features=1)
    ctx.render(2, ptk.SEED_REFERENCE, PIN_SEED)
    assert _same(a, ctx.read_planes()).all()


def test_bad_scenes_are_rejected(ctx):
    import ptk
    import scenes
    sc = scenes.random_scene(40, 32, 16, seed=5)
    bad = dict(sc)
    bad["shape_mat"] = sc["shape_mat"].copy()
    bad["shape_mat"][3] = 99
    with pytest.raises(ptk.PtkError, match="material"):
        ctx.set_scene(bad)
    bad = dict(sc)
    bad["important_shapes"] = np.asarray([40], np.uint32)
    with pytest.raises(ptk.PtkError, match="important"):
        ctx.set_scene(bad)
    bad = dict(sc)
    bad["shape_attrs"] = sc["shape_attrs"].copy()
    box = int(np.nonzero(sc["shape_type"] == 1)[0][0])
    bad["shape_attrs"][box, 9:13] = [2.0, 0.0, 0.0, 0.0]
    with pytest.raises(ptk.PtkError, match="quaternion"):
        ctx.set_scene(bad)
    bad["shape_attrs"][box, 9:13] = [1.0, 0.0, 0.0, 0.0]
    bad["shape_attrs"][box, 0] = np.nan
    with pytest.raises(ptk.PtkError, match="finite"):
        ctx.set_scene(bad)
    bad = dict(sc)
    bad["max_depth"] = 13
    with pytest.raises(ptk.PtkError, match="max_depth"):
        ctx.set_scene(bad)


def test_scenes_bent_towards_what_could_make_a_hierarchy_skip_a_shape(ctx, oracle):
    """scripts/gpu_bvh_fuzz.py: duplicated and nested shapes, everything in one plane, scenes in
    millimetres, in hundreds of metres and 5000 units from the origin, clusters, flat and rotated
    boxes, sizes over four orders of magnitude, box faces in the camera's planes with the jitter
    off; depth 1..12, glass, blur.  Eighteen of them here (two of each kind): the hierarchy, the
    walk that visits everything and the CPU's index-order scan give the same bits in all 38
    planes and the same ray counts."""
    import os
    import sys
    import ptk
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    import gpu_bvh_fuzz as F
    kinds = {}
    seed = 70000
    while len(kinds) < 9 or min(kinds.values()) < 2:
        sc, what = F.bent_scene(seed)
        seed += 1
        if kinds.get(what["kind"], 0) >= 2 or what["n"] > 1500:
            continue
        kinds[what["kind"]] = kinds.get(what["kind"], 0) + 1
        ctx.set_scene(sc, want_features=1)
        rays_a, _ = ctx.render(2, ptk.SEED_REFERENCE, F.SEED)
        a = ctx.read_planes()
        ctx.set_scene(sc, want_features=1, flags=ptk.SCENE_VISIT_ALL)
        rays_b, _ = ctx.render(2, ptk.SEED_REFERENCE, F.SEED)
        assert rays_a == rays_b and _same(a, ctx.read_planes()).all(), what
        with oracle.variant("R"):
            acc = oracle.new_accumulators(sc["res_h"], sc["res_v"])
            _, glo, r1 = oracle.render_scene(sc, acc, 1, 0, F.SEED, want_features=True, threads=8)
            _, _, r2 = oracle.render_scene(sc, acc, 1, 1, glo, want_features=True, threads=8)
        assert rays_a == r1 + r2 and _same(a, acc).all(), what
