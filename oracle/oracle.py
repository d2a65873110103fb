"""ctypes binding for the CPU oracle (oracle/ptk_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The shipped library never touches this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libptk_oracle.so")

N_PLANES = 38
PL_COL, PL_NORMAL, PL_ALB1, PL_ALB2, PL_WPOS, PL_DL, PL_SDV, PL_SD = 0, 3, 6, 9, 12, 15, 16, 32


class Constants(C.Structure):
    """Byte layout of the reference's struct Constants (headers/GLOBALS.h:80-94)."""
    _fields_ = [
        ("cam_pos", C.c_float * 3), ("cam_fwd", C.c_float * 3),
        ("cam_right", C.c_float * 3), ("cam_up", C.c_float * 3),
        ("max_angle", C.c_float), ("rand_samp", C.c_float),
        ("num_shapes", C.c_int32), ("max_depth", C.c_int32),
        ("res_v", C.c_int32), ("res_h", C.c_int32),
        ("max_angle_v", C.c_float), ("max_angle_h", C.c_float), ("focal_length", C.c_float),
        ("shapes", (C.c_int32 * 3) * 50),
        ("attrs", C.c_float * 450),
        ("mats", (C.c_float * 6) * 50),
        ("imp", C.c_uint32 * 10),
        ("num_imp", C.c_uint32),
        ("want_features", C.c_uint32),
    ]


assert C.sizeof(Constants) == 3732


class SceneDesc(C.Structure):
    """orc_scene_desc (same layout as ptk_scene_desc of include/ptk.h)."""
    _fields_ = [("cam_pos", C.c_float * 3), ("cam_forward", C.c_float * 3),
                ("cam_right", C.c_float * 3), ("cam_up", C.c_float * 3),
                ("rand_samp", C.c_float), ("max_depth", C.c_int32),
                ("res_v", C.c_int32), ("res_h", C.c_int32),
                ("max_angle_v", C.c_float), ("max_angle_h", C.c_float),
                ("focal_length", C.c_float),
                ("num_shapes", C.c_int32), ("num_mats", C.c_int32),
                ("num_important_shapes", C.c_int32), ("flags", C.c_int32),
                ("shape_type", C.POINTER(C.c_int32)), ("shape_mat", C.POINTER(C.c_int32)),
                ("shape_attrs", C.POINTER(C.c_float)), ("mat_list", C.POINTER(C.c_float)),
                ("important_shapes", C.POINTER(C.c_uint32))]


class Sample(C.Structure):
    _fields_ = [
        ("xyz", C.c_float * 3), ("normal", C.c_float * 3), ("albedo1", C.c_float * 3),
        ("albedo2", C.c_float * 3), ("world_pos", C.c_float * 3),
        ("direct_light", C.c_float), ("rays_sent", C.c_uint32),
    ]


assert C.sizeof(Sample) == 68


class Scene(C.Structure):
    _fields_ = [
        ("n_mats", C.c_int), ("n_shapes", C.c_int), ("n_imp", C.c_int),
        ("mats", (C.c_float * 6) * 64),
        ("shape_type", C.c_int * 64), ("shape_mat", C.c_int * 64),
        ("shape_pos", (C.c_float * 3) * 64),
        ("sphere_r", C.c_float * 64),
        ("box_min", (C.c_float * 3) * 64), ("box_max", (C.c_float * 3) * 64),
        ("box_rot", (C.c_float * 3) * 64), ("box_q", (C.c_float * 4) * 64),
        ("imp", C.c_int * 64),
        ("cam_pos", C.c_float * 3), ("cam_fwd", C.c_float * 3),
        ("cam_up", C.c_float * 3), ("cam_right", C.c_float * 3),
        ("vfov", C.c_float), ("hfov", C.c_float), ("focal_len", C.c_float),
        ("exposure", C.c_float), ("gamma", C.c_float),
    ]


_VARIANT_LIBS = {"F": "libptk_oracle.so", "D": "libptk_oracle_D.so", "R": "libptk_oracle_R.so"}


def build(force=False, variant="F"):
    """Compile one variant of the oracle with gcc (a few seconds).

    F: unqualified libm calls bind to the float overloads (canonical).
    D: they bind to the C double functions (what a g++/<cmath> build does).
    R: as F with cos/sin/acos/pow correctly rounded (evaluated in double, rounded once).
    """
    name = _VARIANT_LIBS[variant]
    path = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "ptk_oracle.c")
    hdr = os.path.join(_HERE, "ptk_oracle.h")
    if (not force and os.path.exists(path)
            and os.path.getmtime(path) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return path
    subprocess.check_call(["make", "-C", _HERE, "-B", name], stdout=subprocess.DEVNULL)
    return path


_libs = {}
_variant = "F"


class variant:
    """Context manager: route every call of this module to another libm variant of the
    oracle ("F", "D" or "R"; see build())."""

    def __init__(self, name):
        assert name in _VARIANT_LIBS
        self.name = name

    def __enter__(self):
        global _variant
        self.prev, _variant = _variant, self.name
        lib()
        return self

    def __exit__(self, *exc):
        global _variant
        _variant = self.prev
        return False


def lib():
    if _variant not in _libs:
        _libs[_variant] = _bind(C.CDLL(build(variant=_variant)))
    return _libs[_variant]


def _bind(L):
    if True:
        fp = C.POINTER(C.c_float)
        L.orc_load_scene.argtypes = [C.c_char_p, C.POINTER(Scene)]
        L.orc_load_scene.restype = C.c_int
        L.orc_build_constants.argtypes = [C.POINTER(Scene), C.c_int, C.c_int, C.c_int,
                                          C.c_float, C.c_int, C.POINTER(Constants)]
        L.orc_build_constants.restype = None
        L.orc_seed_stream.argtypes = [C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_int64)]
        L.orc_seed_stream.restype = None
        L.orc_rand_between.argtypes = [C.POINTER(C.c_int64), C.c_float, C.c_float]
        L.orc_rand_between.restype = C.c_float
        L.orc_render_sample.argtypes = [C.POINTER(Constants), C.c_int, C.c_int, C.c_int64,
                                        C.c_int64, C.POINTER(Sample)]
        L.orc_render_sample.restype = None
        L.orc_render.argtypes = [C.POINTER(Constants), fp, C.c_int, C.POINTER(C.c_int),
                                 C.POINTER(C.c_uint64), C.c_int, C.POINTER(C.c_uint64)]
        L.orc_render.restype = None
        L.orc_render_scene.argtypes = [C.POINTER(SceneDesc), C.c_int, fp, C.c_int,
                                       C.POINTER(C.c_int), C.POINTER(C.c_uint64), C.c_int,
                                       C.POINTER(C.c_uint64)]
        L.orc_render_scene.restype = C.c_int
        L.orc_render_scene_pixels.argtypes = [C.POINTER(SceneDesc), C.c_int, C.c_int,
                                              C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                              C.POINTER(C.c_int64), fp, C.c_int]
        L.orc_render_scene_pixels.restype = C.c_int
        L.orc_auto_exposure.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.POINTER(C.c_double)]
        L.orc_auto_exposure.restype = C.c_float
        L.orc_forward_prop.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp, fp, fp, fp, fp, C.c_int]
        L.orc_forward_prop.restype = None
        L.orc_filter.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, C.c_int]
        L.orc_filter.restype = None
        L.orc_filter_variant.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp,
                                         C.c_int]
        L.orc_filter_variant.restype = None
        L.orc_mad_procedure.argtypes = [fp, C.c_size_t, fp]
        L.orc_mad_procedure.restype = None
        L.orc_load_weights.argtypes = [C.c_char_p, fp, fp, fp]
        L.orc_load_weights.restype = C.c_int
        L.orc_tonemap8.argtypes = [fp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                   C.POINTER(C.c_uint8)]
        L.orc_tonemap8.restype = None
        L.orc_forward_prop_layers.argtypes = [fp, C.c_int, C.c_int, C.c_int, fp, fp, fp, fp, fp, fp,
                                              fp, C.c_int]
        L.orc_forward_prop_layers.restype = None
        L.orc_rel_mse.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int]
        L.orc_rel_mse.restype = C.c_float
        L.orc_back_prop.argtypes = [fp] * 8 + [C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, fp, fp,
                                               fp]
        L.orc_back_prop.restype = C.c_int
        L.orc_post_process.argtypes = [fp, fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                       C.c_float, fp]
        L.orc_post_process.restype = C.c_int
        L.orc_quantise8.argtypes = [fp, C.c_int, C.c_int, C.POINTER(C.c_uint8)]
        L.orc_quantise8.restype = None
    return L


def _fp(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


def load_scene(path):
    s = Scene()
    rc = lib().orc_load_scene(os.fsencode(path), C.byref(s))
    if rc != 0:
        raise IOError("cannot load scene %s (rc=%d)" % (path, rc))
    return s


def build_constants(scene, width, height, max_depth=4, rand_samp=0.005, denoising=True):
    c = Constants()
    lib().orc_build_constants(C.byref(scene), width, height, max_depth,
                              C.c_float(rand_samp), int(bool(denoising)), C.byref(c))
    return c


def seed_stream(glo, n):
    """Advance the global seed state n steps; returns (n,2) int64 pixel seeds and the new state."""
    st = (C.c_uint64 * 2)(glo[0], glo[1])
    out = np.empty((n, 2), dtype=np.int64)
    lib().orc_seed_stream(st, n, out.ctypes.data_as(C.POINTER(C.c_int64)))
    return out, (int(st[0]), int(st[1]))


def rand_stream(s1, s2, count, lo=0.0, hi=1.0):
    """count successive randBetween(lo, hi) draws from one pixel seed."""
    st = (C.c_int64 * 2)(s1, s2)
    out = np.empty(count, dtype=np.float32)
    f = lib().orc_rand_between
    for i in range(count):
        out[i] = f(st, C.c_float(lo), C.c_float(hi))
    return out, (int(st[0]), int(st[1]))


def render_sample(consts, row, col, s1, s2):
    s = Sample()
    lib().orc_render_sample(C.byref(consts), row, col, s1, s2, C.byref(s))
    return s


def new_accumulators(width, height):
    return np.zeros((N_PLANES, height * width), dtype=np.float32)


def render(consts, acc, spp, sample_count, glo, threads=1):
    """Run spp accumulate passes.  Returns (sample_count, glo, rays_last_pass)."""
    sc = C.c_int(sample_count)
    st = (C.c_uint64 * 2)(glo[0], glo[1])
    rays = C.c_uint64(0)
    lib().orc_render(C.byref(consts), _fp(acc), spp, C.byref(sc), st, threads, C.byref(rays))
    return sc.value, (int(st[0]), int(st[1])), int(rays.value)


def _scene_desc(scene):
    """Scene dict -> (orc_scene_desc, arrays it points into)."""
    d = SceneDesc()
    for k in ("cam_pos", "cam_forward", "cam_right", "cam_up"):
        getattr(d, k)[:] = [float(v) for v in scene[k]]
    for k in ("rand_samp", "max_angle_v", "max_angle_h", "focal_length"):
        setattr(d, k, float(scene[k]))
    for k in ("max_depth", "res_v", "res_h"):
        setattr(d, k, int(scene[k]))
    st = np.ascontiguousarray(scene["shape_type"], dtype=np.int32)
    sm = np.ascontiguousarray(scene["shape_mat"], dtype=np.int32)
    sa = np.ascontiguousarray(scene["shape_attrs"], dtype=np.float32).reshape(-1, 13)
    ml = np.ascontiguousarray(scene["mat_list"], dtype=np.float32).reshape(-1, 6)
    im = np.ascontiguousarray(scene["important_shapes"], dtype=np.uint32)
    d.num_shapes, d.num_mats, d.num_important_shapes, d.flags = len(st), len(ml), len(im), 0
    d.shape_type = st.ctypes.data_as(C.POINTER(C.c_int32))
    d.shape_mat = sm.ctypes.data_as(C.POINTER(C.c_int32))
    d.shape_attrs = sa.ctypes.data_as(C.POINTER(C.c_float))
    d.mat_list = ml.ctypes.data_as(C.POINTER(C.c_float))
    d.important_shapes = im.ctypes.data_as(C.POINTER(C.c_uint32))
    return d, (st, sm, sa, ml, im)


def render_scene(scene, acc, spp, sample_count, glo, want_features=False, threads=1):
    """As render(), over a scene dict with tables of any length: cam_pos, cam_forward,
    cam_right, cam_up, rand_samp, max_depth, res_v, res_h, max_angle_v, max_angle_h,
    focal_length, shape_type [N], shape_mat [N], shape_attrs [N,13], mat_list [M,6],
    important_shapes [K]."""
    d, keep = _scene_desc(scene)
    sc = C.c_int(sample_count)
    state = (C.c_uint64 * 2)(glo[0], glo[1])
    rays = C.c_uint64(0)
    rc = lib().orc_render_scene(C.byref(d), 1 if want_features else 0, _fp(acc), spp, C.byref(sc),
                                state, threads, C.byref(rays))
    assert rc == 0
    return sc.value, (int(state[0]), int(state[1])), int(rays.value)


def render_scene_pixels(scene, rows, cols, seeds, want_features=False, threads=1):
    """One sample of each listed pixel: seeds (n, 2) int64.  Returns (n, 17) float32: colour,
    normal, albedo1, albedo2, worldPos, directLight, rays."""
    d, keep = _scene_desc(scene)
    rows = np.ascontiguousarray(rows, dtype=np.int32)
    cols = np.ascontiguousarray(cols, dtype=np.int32)
    seeds = np.ascontiguousarray(seeds, dtype=np.int64).reshape(-1, 2)
    assert len(rows) == len(cols) == len(seeds)
    out = np.zeros((len(rows), 17), np.float32)
    i32 = C.POINTER(C.c_int32)
    rc = lib().orc_render_scene_pixels(C.byref(d), 1 if want_features else 0, len(rows),
                                       rows.ctypes.data_as(i32), cols.ctypes.data_as(i32),
                                       seeds.ctypes.data_as(C.POINTER(C.c_int64)), _fp(out), threads)
    assert rc == 0
    return out


def auto_exposure(acc, width, height, sample_count, formula=0):
    raw = C.c_double(0)
    e = lib().orc_auto_exposure(_fp(acc), width, height, sample_count, formula, C.byref(raw))
    return float(e), float(raw.value)


def load_weights(path):
    w12 = np.zeros(360, np.float32)
    w23 = np.zeros(100, np.float32)
    w34 = np.zeros(70, np.float32)
    rc = lib().orc_load_weights(os.fsencode(path), _fp(w12), _fp(w23), _fp(w34))
    if rc != 0:
        raise IOError("cannot load weights %s (rc=%d)" % (path, rc))
    return w12, w23, w34


def mad_procedure(windows):
    """The "median getter" procedure on each row of an (n, 9) array."""
    w = np.ascontiguousarray(windows, dtype=np.float32).reshape(-1, 9)
    out = np.empty(len(w), np.float32)
    lib().orc_mad_procedure(_fp(w), len(w), _fp(out))
    return out


def forward_prop(acc, width, height, sample_count, weights, threads=1):
    P = width * height
    feats = np.zeros((P, 36), np.float32)
    var = np.zeros((P, 7), np.float32)
    w12, w23, w34 = weights
    lib().orc_forward_prop(_fp(acc), width, height, sample_count, _fp(w12), _fp(w23), _fp(w34),
                           _fp(feats), _fp(var), threads)
    return feats, var


def filter(acc, variances, width, height, sample_count, n=1, threads=1, colour_variant=0):
    P = width * height
    den = np.zeros((P, 3), np.float32)
    wc = np.zeros(P, np.float32)
    lib().orc_filter_variant(_fp(acc), _fp(variances), width, height, sample_count, n,
                             colour_variant, _fp(den), _fp(wc), threads)
    return den, wc


def forward_prop_layers(acc, width, height, sample_count, weights, threads=1):
    """Returns (features (P,36), variances (P,7), layer2 (P,10), layer3 (P,10))."""
    P = width * height
    w12, w23, w34 = (np.ascontiguousarray(w, dtype=np.float32) for w in weights)
    feats, var = np.empty((P, 36), np.float32), np.empty((P, 7), np.float32)
    l2, l3 = np.empty((P, 10), np.float32), np.empty((P, 10), np.float32)
    lib().orc_forward_prop_layers(_fp(acc), width, height, sample_count, _fp(w12), _fp(w23),
                                  _fp(w34), _fp(feats), _fp(var), _fp(l2), _fp(l3), threads)
    return feats, var, l2, l3


def rel_mse(denoised, target, width, height, sample_count):
    d = np.ascontiguousarray(denoised, dtype=np.float32)
    t = np.ascontiguousarray(target, dtype=np.float32)
    return float(lib().orc_rel_mse(_fp(d), _fp(t), width, height, sample_count))


def back_prop(acc, denoised, wc_sum, variances, feats, l2, l3, target, width, height,
              sample_count, n, learning_rate, weights):
    """CPUBackProp: returns the updated (w12, w23, w34)."""
    arrs = [np.ascontiguousarray(a, dtype=np.float32)
            for a in (acc, denoised, wc_sum, variances, feats, l2, l3, target)]
    w12, w23, w34 = (np.array(w, dtype=np.float32, copy=True) for w in weights)
    rc = lib().orc_back_prop(*[_fp(a) for a in arrs], width, height, sample_count, n,
                             C.c_float(learning_rate), _fp(w12), _fp(w23), _fp(w34))
    if rc != 0:
        raise ValueError("back-prop is undefined for images taller than wide")
    return w12, w23, w34


def post_process(acc, denoised, width, height, sample_count, metric, exposure, gamma):
    """CUDAPostProcess: returns the interleaved (P, 3) post screen."""
    acc = np.ascontiguousarray(acc, dtype=np.float32)
    den = None if denoised is None else np.ascontiguousarray(denoised, dtype=np.float32)
    out = np.empty((width * height, 3), np.float32)
    rc = lib().orc_post_process(_fp(acc), _fp(den) if den is not None else None, width, height,
                                sample_count, metric, C.c_float(exposure), C.c_float(gamma),
                                _fp(out))
    if rc != 0:
        raise ValueError("unsupported display metric %d" % metric)
    return out


def quantise8(post, width, height):
    post = np.ascontiguousarray(post, dtype=np.float32)
    out = np.empty((height, width, 3), np.uint8)
    lib().orc_quantise8(_fp(post), width, height, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def tonemap8(acc, width, height, sample_count, exposure, gamma):
    out = np.zeros((height, width, 3), np.uint8)
    lib().orc_tonemap8(_fp(acc), width, height, sample_count, C.c_float(exposure),
                       C.c_float(gamma), out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out
