"""ctypes binding of libptk.so (the C ABI of include/ptk.h).

This module only marshals arguments; every computation happens in the HIP
library.  There is no CPU fallback: if libptk.so is missing or a call fails,
an exception is raised.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PTK_LIB") or os.path.join(_HERE, "libptk.so")

CONSTANTS_BYTES = 3732
N_PLANES = 38
PL_COL, PL_NORMAL, PL_ALB1, PL_ALB2, PL_WPOS, PL_DL, PL_SDV, PL_SD = 0, 3, 6, 9, 12, 15, 16, 32
SEED_REFERENCE, SEED_COUNTER = 0, 1
MATH_EXACT, MATH_FAST = 0, 1
EXPOSURE_CPU, EXPOSURE_SKEPU, EXPOSURE_MEAN = 0, 1, 2

EXPORTS = [
    "ptk_create", "ptk_destroy", "ptk_last_error", "ptk_set_constants", "ptk_reset_accumulators",
    "ptk_set_math_mode", "ptk_set_stream_id", "ptk_render", "ptk_read_accumulators",
    "ptk_read_planes", "ptk_write_planes", "ptk_planes_device", "ptk_get_sample_count",
    "ptk_set_sample_count", "ptk_refresh_std_dev", "ptk_auto_exposure", "ptk_denoise",
    "ptk_denoise_rows", "ptk_denoise_outputs_device", "ptk_post_process", "ptk_save_target",
    "ptk_set_target", "ptk_train_step", "ptk_timing", "ptk_synchronize", "ptk_set_scene",
    "ptk_check_scene", "ptk_set_camera", "ptk_auto_exposure_static", "ptk_band_layout",
    "ptk_pack_bands", "ptk_merge_received",
]
SCENE_VISIT_ALL = 1
SCENE_ATTRS = 13


class PtkError(RuntimeError):
    pass


class Times(C.Structure):
    _fields_ = [("render_ms", C.c_double), ("seed_ms", C.c_double), ("exposure_ms", C.c_double),
                ("features_ms", C.c_double), ("filter_ms", C.c_double), ("denoise_ms", C.c_double),
                ("render_samples", C.c_uint64), ("render_rays", C.c_uint64),
                ("post_ms", C.c_double), ("train_ms", C.c_double)]


class SceneDesc(C.Structure):
    """ptk_scene_desc."""
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


class SceneStats(C.Structure):
    _fields_ = [("num_shapes", C.c_int32), ("nodes", C.c_int32), ("leaves", C.c_int32),
                ("depth", C.c_int32), ("build_ms", C.c_double), ("device_bytes", C.c_uint64)]


def scene_desc(scene, flags=0, cls=SceneDesc):
    """Marshals a scene dict into a ptk_scene_desc.  Keys: cam_pos, cam_forward, cam_right,
    cam_up, rand_samp, max_depth, res_v, res_h, max_angle_v, max_angle_h, focal_length,
    shape_type [N], shape_mat [N], shape_attrs [N,13], mat_list [M,6], important_shapes [K].
    Returns (struct, arrays the struct points into -- keep them alive)."""
    d = cls()
    for k in ("cam_pos", "cam_forward", "cam_right", "cam_up"):
        getattr(d, k)[:] = [float(v) for v in scene[k]]
    for k in ("rand_samp", "max_angle_v", "max_angle_h", "focal_length"):
        setattr(d, k, float(scene[k]))
    for k in ("max_depth", "res_v", "res_h"):
        setattr(d, k, int(scene[k]))
    st = np.ascontiguousarray(scene["shape_type"], dtype=np.int32)
    sm = np.ascontiguousarray(scene["shape_mat"], dtype=np.int32)
    sa = np.ascontiguousarray(scene["shape_attrs"], dtype=np.float32).reshape(-1, SCENE_ATTRS)
    ml = np.ascontiguousarray(scene["mat_list"], dtype=np.float32).reshape(-1, 6)
    im = np.ascontiguousarray(scene["important_shapes"], dtype=np.uint32)
    assert len(st) == len(sm) == len(sa)
    d.num_shapes, d.num_mats, d.num_important_shapes, d.flags = len(st), len(ml), len(im), flags
    d.shape_type = st.ctypes.data_as(C.POINTER(C.c_int32))
    d.shape_mat = sm.ctypes.data_as(C.POINTER(C.c_int32))
    d.shape_attrs = sa.ctypes.data_as(C.POINTER(C.c_float))
    d.mat_list = ml.ctypes.data_as(C.POINTER(C.c_float))
    d.important_shapes = im.ctypes.data_as(C.POINTER(C.c_uint32))
    return d, (st, sm, sa, ml, im)


def scene_from_constants(constants):
    """The 3732-byte scene POD as a scene dict (for ptk_set_scene)."""
    raw = bytes(constants)
    f = np.frombuffer(raw, dtype=np.float32)
    i = np.frombuffer(raw, dtype=np.int32)
    u = np.frombuffer(raw, dtype=np.uint32)
    n = int(i[14])
    shapes = i[21:21 + 150].reshape(50, 3)
    attrs = f[171:171 + 450]
    sa = np.zeros((n, SCENE_ATTRS), np.float32)
    for s in range(n):
        a = int(shapes[s, 2])
        k = 4 if shapes[s, 0] == 0 else 13
        sa[s, :k] = attrs[a:a + k]
    num_imp = int(u[931])
    return dict(cam_pos=f[0:3], cam_forward=f[3:6], cam_right=f[6:9], cam_up=f[9:12],
                rand_samp=f[13], max_depth=int(i[15]), res_v=int(i[16]), res_h=int(i[17]),
                max_angle_v=f[18], max_angle_h=f[19], focal_length=f[20],
                shape_type=shapes[:n, 0].copy(), shape_mat=shapes[:n, 1].copy(), shape_attrs=sa,
                mat_list=f[621:621 + 300].reshape(50, 6).copy(),
                important_shapes=u[921:921 + num_imp].copy())


_lib = None


def load():
    """Loads libptk.so; raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PtkError("libptk.so not found at %s: build it with `make -C path-tracer_amd` "
                       "(there is no CPU fallback)" % LIB_PATH)
    # PyTorch wheels bundle their own HIP runtime under the same soname as the system's.
    # Whichever copy a process loads first serves every later user, and torch finds no GPU
    # through the system copy: when torch is present it must be loaded before libptk.so.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    fp = C.POINTER(C.c_float)
    vp = C.c_void_p
    L.ptk_last_error.restype = C.c_char_p
    L.ptk_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.ptk_destroy.argtypes = [vp]
    L.ptk_set_constants.argtypes = [vp, vp, C.c_size_t, C.c_int]
    L.ptk_reset_accumulators.argtypes = [vp]
    L.ptk_set_math_mode.argtypes = [vp, C.c_int]
    L.ptk_set_stream_id.argtypes = [vp, C.c_uint64]
    L.ptk_render.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.ptk_read_accumulators.argtypes = [vp, fp, fp, fp, fp, fp, fp, fp, C.POINTER(C.c_int)]
    L.ptk_read_planes.argtypes = [vp, C.c_int, C.c_int, fp]
    L.ptk_write_planes.argtypes = [vp, C.c_int, C.c_int, fp, C.c_int]
    L.ptk_planes_device.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.ptk_get_sample_count.argtypes = [vp, C.POINTER(C.c_int)]
    L.ptk_set_sample_count.argtypes = [vp, C.c_int]
    L.ptk_refresh_std_dev.argtypes = [vp]
    L.ptk_auto_exposure.argtypes = [vp, C.c_int, fp, C.POINTER(C.c_double)]
    L.ptk_denoise.argtypes = [vp, fp, fp, fp, C.c_int, C.c_int, fp, fp, fp, fp]
    L.ptk_denoise_rows.argtypes = [vp, fp, fp, fp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.ptk_denoise_outputs_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.ptk_post_process.argtypes = [vp, C.c_int, C.c_float, C.c_float, fp, C.POINTER(C.c_uint8)]
    L.ptk_save_target.argtypes = [vp]
    L.ptk_set_target.argtypes = [vp, fp]
    L.ptk_train_step.argtypes = [vp, fp, fp, fp, C.c_int, C.c_float, fp]
    L.ptk_set_scene.argtypes = [vp, C.POINTER(SceneDesc), C.c_int, C.POINTER(SceneStats)]
    L.ptk_set_camera.argtypes = [vp, fp, fp, fp, fp]
    L.ptk_check_scene.argtypes = [C.POINTER(SceneDesc), C.POINTER(SceneStats)]
    L.ptk_auto_exposure_static.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, vp,
                                           vp, vp]
    ip = C.POINTER(C.c_int)
    L.ptk_band_layout.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, ip, ip, ip]
    L.ptk_pack_bands.argtypes = [vp, C.c_int, C.c_int, vp, C.c_size_t]
    L.ptk_merge_received.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, C.c_size_t, ip]
    L.ptk_timing.argtypes = [vp, C.POINTER(Times)]
    L.ptk_synchronize.argtypes = [vp]
    for name in EXPORTS:
        if name != "ptk_last_error":
            getattr(L, name).restype = C.c_int
    _lib = L
    return L


def band_layout(height, world, rank, halo):
    """(row_lo, row_hi, rows_max) of ptk_band_layout."""
    lo, hi, most = C.c_int(), C.c_int(), C.c_int()
    rc = load().ptk_band_layout(height, world, rank, halo, C.byref(lo), C.byref(hi), C.byref(most))
    if rc != 0:
        raise PtkError(load().ptk_last_error().decode())
    return lo.value, hi.value, most.value


def _fp(a):
    if a is None:
        return None
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_float))


def check_scene(scene, flags=0):
    """ptk_check_scene: validates a scene dict and verifies the hierarchy built for it, on the
    host alone.  Returns the statistics as a dict; raises PtkError on a bad scene."""
    L = load()
    d, keep = scene_desc(scene, flags)
    st = SceneStats()
    rc = L.ptk_check_scene(C.byref(d), C.byref(st))
    del keep
    if rc != 0:
        raise PtkError("libptk error %d: %s" % (rc, L.ptk_last_error().decode()))
    return {k: getattr(st, k) for k, _ in SceneStats._fields_}


class Context:
    """One GPU's renderer/denoiser state (wraps ptk_ctx)."""

    def __init__(self, device=0):
        self._L = load()
        self._h = C.c_void_p()
        self._check(self._L.ptk_create(device, C.byref(self._h)))
        self.width = self.height = 0

    def _check(self, rc):
        if rc != 0:
            raise PtkError("libptk error %d: %s" % (rc, self._L.ptk_last_error().decode()))

    def close(self):
        if self._h:
            self._L.ptk_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_constants(self, constants, want_features=-1):
        """constants: any object exposing the 3732-byte scene POD through the buffer protocol."""
        raw = bytes(constants)
        buf = C.create_string_buffer(raw, len(raw))
        self._check(self._L.ptk_set_constants(self._h, C.cast(buf, C.c_void_p), len(raw),
                                              want_features))
        hdr = np.frombuffer(raw, dtype=np.int32, count=20)
        self.height, self.width = int(hdr[16]), int(hdr[17])

    def set_scene(self, scene, want_features=0, flags=0):
        """scene: dict as for scene_desc().  Returns the hierarchy's statistics as a dict."""
        d, keep = scene_desc(scene, flags)
        st = SceneStats()
        self._check(self._L.ptk_set_scene(self._h, C.byref(d), int(want_features), C.byref(st)))
        del keep
        self.height, self.width = int(d.res_v), int(d.res_h)
        return {k: getattr(st, k) for k, _ in SceneStats._fields_}

    def set_camera(self, pos, forward, right, up):
        v = [np.ascontiguousarray(x, dtype=np.float32) for x in (pos, forward, right, up)]
        self._check(self._L.ptk_set_camera(self._h, *[_fp(x) for x in v]))

    @property
    def pixels(self):
        return self.width * self.height

    def reset(self):
        self._check(self._L.ptk_reset_accumulators(self._h))

    def set_math_mode(self, mode):
        self._check(self._L.ptk_set_math_mode(self._h, mode))

    def set_stream_id(self, sid):
        self._check(self._L.ptk_set_stream_id(self._h, sid))

    def render(self, spp, seed_mode=SEED_COUNTER, glo=None):
        """Returns (rays, glo_after)."""
        rays = C.c_uint64(0)
        if seed_mode == SEED_REFERENCE:
            st = (C.c_uint64 * 2)(glo[0], glo[1])
            self._check(self._L.ptk_render(self._h, spp, seed_mode, st, C.byref(rays)))
            return int(rays.value), (int(st[0]), int(st[1]))
        self._check(self._L.ptk_render(self._h, spp, seed_mode, None, C.byref(rays)))
        return int(rays.value), glo

    def read_planes(self, first=0, count=N_PLANES):
        out = np.empty((count, self.pixels), dtype=np.float32)
        self._check(self._L.ptk_read_planes(self._h, first, count, _fp(out)))
        return out

    def write_planes(self, planes, sample_count, first=0):
        planes = np.ascontiguousarray(planes, dtype=np.float32)
        self._check(self._L.ptk_write_planes(self._h, first, planes.shape[0], _fp(planes),
                                             sample_count))

    def read_accumulators(self):
        """The accumulators in the layout of the reference's globals."""
        P = self.pixels
        out = {k: np.empty((P, 3), np.float32) for k in
               ("preScreen", "normal", "albedo1", "albedo2", "worldPos")}
        out["directLight"] = np.empty(P, np.float32)
        out["stdDev"] = np.empty((P, 6), np.float32)
        sc = C.c_int(0)
        self._check(self._L.ptk_read_accumulators(
            self._h, _fp(out["preScreen"]), _fp(out["normal"]), _fp(out["albedo1"]),
            _fp(out["albedo2"]), _fp(out["directLight"]), _fp(out["worldPos"]),
            _fp(out["stdDev"]), C.byref(sc)))
        out["sampleCount"] = sc.value
        return out

    def planes_device(self):
        base = C.c_void_p()
        n = C.c_size_t(0)
        self._check(self._L.ptk_planes_device(self._h, C.byref(base), C.byref(n)))
        return int(base.value), int(n.value)

    @property
    def sample_count(self):
        sc = C.c_int(0)
        self._check(self._L.ptk_get_sample_count(self._h, C.byref(sc)))
        return sc.value

    @sample_count.setter
    def sample_count(self, v):
        self._check(self._L.ptk_set_sample_count(self._h, int(v)))

    def refresh_std_dev(self):
        self._check(self._L.ptk_refresh_std_dev(self._h))

    def auto_exposure(self, formula=EXPOSURE_CPU):
        e = C.c_float(0)
        raw = C.c_double(0)
        self._check(self._L.ptk_auto_exposure(self._h, formula, C.byref(e), C.byref(raw)))
        return float(e.value), float(raw.value)

    def denoise(self, weights, n=1, variant=0, want_outputs=True, want_features=False):
        """Returns dict(denoised (P,3), wcSum (P,), variances (P,7), features (P,36)|None)."""
        w12, w23, w34 = (np.ascontiguousarray(w, dtype=np.float32) for w in weights)
        assert w12.size == 360 and w23.size == 100 and w34.size == 70
        P = self.pixels
        res = {"denoised": None, "wcSum": None, "variances": None, "features": None}
        if want_outputs:
            res["denoised"] = np.empty((P, 3), np.float32)
            res["wcSum"] = np.empty(P, np.float32)
            res["variances"] = np.empty((P, 7), np.float32)
        if want_features:
            res["features"] = np.empty((P, 36), np.float32)
        self._check(self._L.ptk_denoise(self._h, _fp(w12), _fp(w23), _fp(w34), n, variant,
                                        _fp(res["denoised"]), _fp(res["wcSum"]),
                                        _fp(res["variances"]), _fp(res["features"])))
        return res

    def denoise_rows(self, weights, n, variant, row_begin, row_end):
        """Denoises rows [row_begin, row_end); results stay on the device."""
        w12, w23, w34 = (np.ascontiguousarray(w, dtype=np.float32) for w in weights)
        self._check(self._L.ptk_denoise_rows(self._h, _fp(w12), _fp(w23), _fp(w34), n, variant,
                                             row_begin, row_end))

    def denoise_outputs_device(self):
        """Device addresses (denoised 3 planes, wcSum 1 plane, variances 7 planes)."""
        d, w, v = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._check(self._L.ptk_denoise_outputs_device(self._h, C.byref(d), C.byref(w),
                                                       C.byref(v)))
        return int(d.value), int(w.value), int(v.value)

    def pack_bands(self, world, halo, send_ptr, send_floats):
        """Fills the device buffer at `send_ptr` with this GPU's planes laid out per destination
        rank (ptk_pack_bands)."""
        self._check(self._L.ptk_pack_bands(self._h, world, halo, C.c_void_p(send_ptr),
                                           C.c_size_t(send_floats)))

    def merge_received(self, world, rank, halo, recv_ptr, recv_floats, samples_per_rank):
        """Merges the blocks received from every rank into the planes of this rank's row span
        (ptk_merge_received); the context's sample count becomes the total."""
        n = (C.c_int * max(1, len(samples_per_rank)))(*[int(x) for x in samples_per_rank])
        self._check(self._L.ptk_merge_received(self._h, world, rank, halo, C.c_void_p(recv_ptr),
                                               C.c_size_t(recv_floats), n))

    def post_process(self, metric=0, exposure=1.0, g=2.2, want_post=True, want_rgb8=False):
        """Returns (postScreen (P, 3) float32 | None, rgb8 (H, W, 3) uint8 top row first | None)."""
        post = np.empty((self.pixels, 3), np.float32) if want_post else None
        rgb = np.empty((self.height, self.width, 3), np.uint8) if want_rgb8 else None
        self._check(self._L.ptk_post_process(
            self._h, metric, C.c_float(exposure), C.c_float(g), _fp(post),
            rgb.ctypes.data_as(C.POINTER(C.c_uint8)) if rgb is not None else None))
        return post, rgb

    def save_target(self):
        self._check(self._L.ptk_save_target(self._h))

    def set_target(self, target):
        t = np.ascontiguousarray(target, dtype=np.float32)
        assert t.size == 3 * self.pixels
        self._check(self._L.ptk_set_target(self._h, _fp(t)))

    def train_step(self, weights, n=1, learning_rate=1e-4):
        """One TrainNN epoch after the render: returns (updated (w12, w23, w34), relMSE)."""
        w12, w23, w34 = (np.array(w, dtype=np.float32, copy=True) for w in weights)
        rel = np.zeros(1, np.float32)
        self._check(self._L.ptk_train_step(self._h, _fp(w12), _fp(w23), _fp(w34), n,
                                           C.c_float(learning_rate), _fp(rel)))
        return (w12, w23, w34), float(rel[0])

    def timing(self):
        t = Times()
        self._check(self._L.ptk_timing(self._h, C.byref(t)))
        return t

    def synchronize(self):
        self._check(self._L.ptk_synchronize(self._h))


# ---- host facade helpers (libptk_host.so: the product's own scene / weight loaders) ----

HOST_LIB_PATH = os.path.join(_HERE, "libptk_host.so")
_host = None


def _load_host():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise PtkError("libptk_host.so not found at %s: build it with "
                           "`make -C path-tracer_amd`" % HOST_LIB_PATH)
        load()  # libptk_host.so depends on libptk.so
        H = C.CDLL(HOST_LIB_PATH)
        H.ptkhost_scene_constants.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                              C.c_int, C.c_void_p, C.POINTER(C.c_float),
                                              C.POINTER(C.c_float)]
        H.ptkhost_scene_constants.restype = C.c_int
        H.ptkhost_load_weights.argtypes = [C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                           C.POINTER(C.c_float)]
        H.ptkhost_load_weights.restype = C.c_int
        for fn in (H.ptkhost_resave_scene, H.ptkhost_resave_weights):
            fn.argtypes = [C.c_char_p, C.c_char_p]
            fn.restype = C.c_int
        _host = H
    return _host


def scene_constants(path, width, height, max_depth=4, rand_samp=0.005, denoising=False):
    """Scene file -> (3732-byte POD, exposure, gamma) via Scene::LoadScene + UpdateConstants."""
    buf = C.create_string_buffer(CONSTANTS_BYTES)
    e, g = C.c_float(0), C.c_float(0)
    rc = _load_host().ptkhost_scene_constants(os.fsencode(path), width, height, max_depth,
                                              C.c_float(rand_samp), int(bool(denoising)), buf,
                                              C.byref(e), C.byref(g))
    if rc != 0:
        raise PtkError("cannot load scene %s" % path)
    return buf.raw, float(e.value), float(g.value)


def scene_tables(path, width, height, max_depth=4, rand_samp=0.005, denoising=False):
    """Scene file of any size -> (scene dict for Context.set_scene, exposure, gamma): the
    tables the host facade hands to ptk_set_scene when a scene exceeds the reference's POD."""
    H = _load_host()
    H.ptkhost_scene_tables.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int,
                                       C.POINTER(SceneDesc), C.POINTER(C.c_int32),
                                       C.POINTER(C.c_int32), C.POINTER(C.c_float),
                                       C.POINTER(C.c_float), C.POINTER(C.c_uint32),
                                       C.POINTER(C.c_float), C.POINTER(C.c_float)]
    H.ptkhost_scene_tables.restype = C.c_int
    d = SceneDesc()
    args = (os.fsencode(path), width, height, max_depth, C.c_float(rand_samp), int(bool(denoising)))
    if H.ptkhost_scene_tables(*args, C.byref(d), None, None, None, None, None, None, None) != 0:
        raise PtkError("cannot load scene %s" % path)
    st = np.zeros(d.num_shapes, np.int32)
    sm = np.zeros(d.num_shapes, np.int32)
    sa = np.zeros((d.num_shapes, SCENE_ATTRS), np.float32)
    ml = np.zeros((d.num_mats, 6), np.float32)
    im = np.zeros(d.num_important_shapes, np.uint32)
    e, g = C.c_float(0), C.c_float(0)
    i32, u32 = C.POINTER(C.c_int32), C.POINTER(C.c_uint32)
    rc = H.ptkhost_scene_tables(*args, C.byref(d), st.ctypes.data_as(i32), sm.ctypes.data_as(i32),
                                _fp(sa), _fp(ml), im.ctypes.data_as(u32), C.byref(e), C.byref(g))
    if rc != 0:
        raise PtkError("cannot load scene %s" % path)
    scene = dict(cam_pos=list(d.cam_pos), cam_forward=list(d.cam_forward),
                 cam_right=list(d.cam_right), cam_up=list(d.cam_up), rand_samp=d.rand_samp,
                 max_depth=d.max_depth, res_v=d.res_v, res_h=d.res_h, max_angle_v=d.max_angle_v,
                 max_angle_h=d.max_angle_h, focal_length=d.focal_length, shape_type=st,
                 shape_mat=sm, shape_attrs=sa, mat_list=ml, important_shapes=im)
    return scene, float(e.value), float(g.value)


def load_weights(path):
    """Weights file -> (w12[360], w23[100], w34[70]) via DenoiserNN::LoadWeights."""
    w12, w23, w34 = np.zeros(360, np.float32), np.zeros(100, np.float32), np.zeros(70, np.float32)
    rc = _load_host().ptkhost_load_weights(os.fsencode(path), _fp(w12), _fp(w23), _fp(w34))
    if rc != 0:
        raise PtkError("cannot load weights %s" % path)
    return w12, w23, w34


def mad_network_host(windows):
    """The denoiser kernel's "median getter" network (csrc/mad_network.h) evaluated on the host
    for each row of an (n, 9) array: the kernel's own code, instantiated with bools."""
    H = _load_host()
    H.ptkhost_mad_network.argtypes = [C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_float)]
    H.ptkhost_mad_network.restype = None
    w = np.ascontiguousarray(windows, dtype=np.float32).reshape(-1, 9)
    out = np.empty(len(w), np.float32)
    H.ptkhost_mad_network(_fp(w), len(w), _fp(out))
    return out


def resave_scene(in_path, out_path):
    """Scene::LoadScene then Scene::SaveScene."""
    if _load_host().ptkhost_resave_scene(os.fsencode(in_path), os.fsencode(out_path)) != 0:
        raise PtkError("cannot re-save scene %s" % in_path)


def resave_weights(in_path, out_path):
    """DenoiserNN::LoadWeights then DenoiserNN::OutputWeights."""
    if _load_host().ptkhost_resave_weights(os.fsencode(in_path), os.fsencode(out_path)) != 0:
        raise PtkError("cannot re-save weights %s" % in_path)
