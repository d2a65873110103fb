"""Synthetic scenes beyond the reference's 50-shape limit, as scene dicts for
Context.set_scene (and for the CPU checker's render_scene, which takes the same dict).

The reference ships no such scene (its POD cannot hold one, headers/GLOBALS.h:80-94);
these generators stand in for them.  Everything is derived from a seed.
"""
import math

import numpy as np


def _camera(width, height, pos, vfov=90.0, hfov=120.0, max_depth=4, rand_samp=0.005):
    # forward -z... no: the reference's default camera looks along +z with y up
    # (PT.cpp:62-72); the angles are the scene file's `c` line.
    return dict(cam_pos=np.asarray(pos, np.float32), cam_forward=np.float32([0, 0, 1]),
                cam_right=np.float32([1, 0, 0]), cam_up=np.float32([0, 1, 0]),
                rand_samp=rand_samp, max_depth=max_depth, res_v=height, res_h=width,
                max_angle_v=math.tan(math.radians(vfov) / 2.0),
                max_angle_h=math.tan(math.radians(hfov) / 2.0), focal_length=1.0)


def _unit_quaternions(rng, n):
    q = rng.normal(size=(n, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    return q.astype(np.float32)


def random_scene(n, width, height, seed=1, dielectric=True, blur=False, max_depth=4,
                 extent=12.0, lights=4, extra_mats=0):
    """n shapes scattered in a room: spheres, boxes (a third of them rotated, a few flat),
    lambert / metal / dielectric materials (plus extra_mats random lambert ones), `lights`
    emissive shapes listed as important."""
    rng = np.random.default_rng(seed)
    mats = [[0.8, 0.8, 0.8, 0, 0, 0], [0.9, 0.3, 0.3, 0, 0, 0], [0.3, 0.9, 0.3, 0, 0, 0],
            [0.3, 0.3, 0.9, 0, 0, 0], [12.0, 12.0, 12.0, 0, 0, 1],
            [0.9, 0.9, 0.9, 0.1 if blur else 0.0, 0, 2]]
    if dielectric:
        mats.append([1.0, 1.0, 1.0, 0.05 if blur else 0.0, 1.5, 3])
    for _ in range(extra_mats):
        mats.append([*rng.uniform(0.2, 0.95, 3), 0, 0, 0])
    mats = np.asarray(mats, np.float32)
    kinds = np.arange(len(mats))
    kinds = kinds[kinds != 4]
    st = np.zeros(n, np.int32)
    sm = np.zeros(n, np.int32)
    sa = np.zeros((n, 13), np.float32)
    # a floor and a back wall so that paths bounce
    fixed = 0
    if n >= 2:
        st[0] = 1
        sa[0, 0:3] = [0, -extent / 2, extent]
        sa[0, 3:6] = [-extent * 2, -extent / 2, -2]
        sa[0, 6:9] = [extent * 2, -extent / 2, extent * 3]
        sa[0, 9] = 1.0
        st[1] = 1
        sa[1, 0:3] = [0, 0, extent * 3]
        sa[1, 3:6] = [-extent * 2, -extent / 2, extent * 3]
        sa[1, 6:9] = [extent * 2, extent * 2, extent * 3]
        sa[1, 9] = 1.0
        fixed = 2
    size = extent / max(2.0, n ** (1.0 / 3.0)) * 0.45
    for s in range(fixed, n):
        c = rng.uniform([-extent, -extent / 2 + size, 2.0], [extent, extent, extent * 2.8])
        sm[s] = rng.choice(kinds)
        if rng.random() < 0.4:
            sa[s, 0:3] = c
            sa[s, 3] = rng.uniform(0.3, 1.0) * size
        else:
            st[s] = 1
            h = rng.uniform(0.2, 1.0, 3) * size
            if rng.random() < 0.1:
                h[rng.integers(3)] = 0.0        # a flat box
            sa[s, 0:3] = c
            sa[s, 3:6] = c - h
            sa[s, 6:9] = c + h
            sa[s, 9] = 1.0
            if rng.random() < 0.35:
                sa[s, 9:13] = _unit_quaternions(rng, 1)[0]
    imp = []
    for k in range(min(lights, max(0, n - fixed))):
        s = fixed + k * max(1, (n - fixed) // max(1, lights))
        sm[s] = 4
        imp.append(s)
    scene = _camera(width, height, [0.0, 1.0, -1.0], max_depth=max_depth)
    scene.update(shape_type=st, shape_mat=sm, shape_attrs=sa, mat_list=mats,
                 important_shapes=np.asarray(imp, np.uint32))
    return scene


def grid_scene(side, width, height, seed=1, max_depth=4):
    """side^3 small shapes on a jittered lattice over a floor, under two sphere lights:
    the scalable benchmark scene (side = 46 gives ~100 k shapes)."""
    rng = np.random.default_rng(seed)
    n = side ** 3 + 3
    mats = np.asarray([[0.75, 0.75, 0.75, 0, 0, 0], [0.85, 0.4, 0.3, 0, 0, 0],
                       [0.3, 0.6, 0.85, 0, 0, 0], [0.9, 0.9, 0.9, 0, 0, 2],
                       [15.0, 15.0, 15.0, 0, 0, 1]], np.float32)
    st = np.zeros(n, np.int32)
    sm = np.zeros(n, np.int32)
    sa = np.zeros((n, 13), np.float32)
    room = 20.0
    st[0] = 1
    sa[0, 0:3] = [0, -room / 2, room]
    sa[0, 3:6] = [-room * 2, -room / 2, -2]
    sa[0, 6:9] = [room * 2, -room / 2, room * 3]
    sa[0, 9] = 1.0
    for k, x in ((1, -room / 2), (2, room / 2)):
        sa[k, 0:3] = [x, room * 0.9, room * 0.6]
        sa[k, 3] = 2.0
        sm[k] = 4
    cell = room * 1.6 / side
    idx = np.stack(np.meshgrid(np.arange(side), np.arange(side), np.arange(side), indexing="ij"),
                   -1).reshape(-1, 3)
    centre = (idx + 0.5) * cell + np.float32([-room * 0.8, -room / 2 + 0.2, room * 0.25])
    centre = centre + rng.uniform(-0.2, 0.2, centre.shape) * cell
    m = side ** 3
    body = slice(3, 3 + m)
    sm[body] = rng.integers(0, 4, m)
    is_box = rng.random(m) < 0.5
    st[body] = is_box.astype(np.int32)
    sa[body, 0:3] = centre
    half = rng.uniform(0.12, 0.3, (m, 3)) * cell
    sa[body, 3] = np.where(is_box, (centre - half)[:, 0], half[:, 0])
    sa[body, 4:6] = np.where(is_box[:, None], (centre - half)[:, 1:3], 0.0)
    sa[body, 6:9] = np.where(is_box[:, None], centre + half, 0.0)
    q = _unit_quaternions(rng, m)
    rotated = is_box & (rng.random(m) < 0.3)
    sa[body, 9] = np.where(is_box, np.where(rotated, q[:, 0], 1.0), 0.0)
    sa[body, 10:13] = np.where(rotated[:, None], q[:, 1:4], 0.0)
    scene = _camera(width, height, [0.0, 2.0, -2.0], max_depth=max_depth)
    scene.update(shape_type=st, shape_mat=sm, shape_attrs=sa, mat_list=mats,
                 important_shapes=np.asarray([1, 2], np.uint32))
    return scene
