"""The headless driver (the reference's frame loop without a window, PT.cpp:105-131) end to
end: scene file -> render -> auto exposure -> denoise -> post-process -> PPM, compared with
the same steps issued through the C ABI from Python."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, scene_path, weights_path

pytestmark = pytest.mark.gpu

HEADLESS = os.path.join(ROOT, "path-tracer_amd", "ptk_headless")


def _read_ppm(path):
    tok = open(path).read().split()
    assert tok[0] == "P3" and tok[3] == "255"
    w, h = int(tok[1]), int(tok[2])
    return np.array(tok[4:], dtype=np.int64).reshape(h, w, 3)


@pytest.mark.parametrize("denoise", [False, True])
def test_headless_ppm_equals_the_abi_pipeline(tmp_path, denoise):
    import ptk
    W, H, spp = 96, 54, 12
    ppm = str(tmp_path / "out.ppm")
    cmd = [HEADLESS, "--scene", scene_path("cornell_crystal"), "--res", "%dx%d" % (W, H),
           "--spp", str(spp), "--counter-seeds", "--auto-exposure", "--ppm", ppm,
           "--dump", str(tmp_path / "d")]
    if denoise:
        cmd += ["--denoise", "1", "--weights", weights_path("BEST")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "Msamples/s" in r.stdout

    raw, _, gamma = ptk.scene_constants(scene_path("cornell_crystal"), W, H, 4, 0.005, denoise)
    ctx = ptk.Context(0)
    ctx.set_constants(raw)
    ctx.render(spp, ptk.SEED_COUNTER)
    exposure, _ = ctx.auto_exposure(ptk.EXPOSURE_CPU)
    if denoise:
        ctx.denoise(ptk.load_weights(weights_path("BEST")), 1, 0, want_outputs=False)
    _, rgb = ctx.post_process(1 if denoise else 0, exposure, gamma, want_post=False,
                              want_rgb8=True)
    pre = ctx.read_accumulators()["preScreen"]
    ctx.close()
    assert np.array_equal(np.fromfile(str(tmp_path / "d.preScreen.f32"), np.float32),
                          pre.reshape(-1))
    got = _read_ppm(ppm)
    # the driver quantises its float screen on the host, the ABI on the device: same rule
    assert np.array_equal(np.clip(got, 0, 255), rgb.astype(np.int64))


def _write_big_scene(path, n, seed=3):
    """A scene file with more shapes than the reference's POD holds (n > 50), in the
    reference's format (headers/Scene.h:226-274)."""
    rng = np.random.default_rng(seed)
    lines = ["m 0.73/0.73/0.73 0 1 0", "m 0.65/0.05/0.05 0 1 0", "m 0.12/0.45/0.15 0 1 0",
             "m 6/6/6 0 1 1", "m 1/1/1 0 1 2", "m 1/1/1 0 1.5 3"]
    lines += ["o 1", "d 0/-6/8 -30/0/-12 30/0/30 0/0/0 0"]          # floor
    lines += ["o 0", "d 0/9/8 2 3"]                                   # light
    for k in range(n - 2):
        c = rng.uniform([-10, -5, 2], [10, 6, 20])
        mat = int(rng.choice([0, 1, 2, 4, 5]))
        if rng.random() < 0.5:
            lines += ["o 0", "d %.4f/%.4f/%.4f %.4f %d" % (c[0], c[1], c[2], rng.uniform(0.2, 0.6), mat)]
        else:
            h = rng.uniform(0.2, 0.7, 3)
            rot = rng.uniform(0, 3.0, 3) if rng.random() < 0.4 else np.zeros(3)
            lines += ["o 1", "d %.4f/%.4f/%.4f %.4f/%.4f/%.4f %.4f/%.4f/%.4f %.3f/%.3f/%.3f %d" % (
                c[0], c[1], c[2], -h[0], -h[1], -h[2], h[0], h[1], h[2], rot[0], rot[1], rot[2], mat)]
    lines += ["i 1", "c 0/0/-9 0/0/1 0/1/0 1/0/0 90 120", "pp 1.5 2.2"]
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def test_headless_scene_beyond_the_pod_goes_through_the_hierarchy(tmp_path):
    """A scene file of 400 shapes: the facade routes it to ptk_set_scene; the frame equals the
    one the ABI gives for the tables the facade built."""
    import ptk
    W, H, spp = 96, 54, 6
    scene_file = str(tmp_path / "big.scene")
    _write_big_scene(scene_file, 400)
    ppm = str(tmp_path / "big.ppm")
    r = subprocess.run([HEADLESS, "--scene", scene_file, "--res", "%dx%d" % (W, H), "--spp", str(spp),
                        "--counter-seeds", "--ppm", ppm, "--dump", str(tmp_path / "d")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    scene, exposure, gamma = ptk.scene_tables(scene_file, W, H)
    assert len(scene["shape_type"]) == 400 and abs(exposure - 1.5) < 1e-6
    ctx = ptk.Context(0)
    ctx.set_scene(scene)
    ctx.render(spp, ptk.SEED_COUNTER)
    _, rgb = ctx.post_process(0, exposure, gamma, want_post=False, want_rgb8=True)
    pre = ctx.read_accumulators()["preScreen"]
    ctx.close()
    assert np.array_equal(np.fromfile(str(tmp_path / "d.preScreen.f32"), np.float32),
                          pre.reshape(-1))
    assert np.array_equal(np.clip(_read_ppm(ppm), 0, 255), rgb.astype(np.int64))
    assert pre.mean() > 0.01


def test_headless_tolerance_held_feature_pass(tmp_path):
    """--denoise-variant 2 selects the feature pass that derives the 7x7 statistics from shared
    column sums; its denoised frame stays within the tolerance of that pass (1e-3 relative on
    the filtered colour) of the frame the reference-arithmetic pass gives, and the variant is
    named on the timing line."""
    W, H, spp = 192, 108, 16
    out = {}
    for variant in (0, 2):
        prefix = str(tmp_path / ("v%d" % variant))
        r = subprocess.run([HEADLESS, "--scene", scene_path("cornell_crystal"), "--res",
                            "%dx%d" % (W, H), "--spp", str(spp), "--counter-seeds", "--denoise", "1",
                            "--weights", weights_path("BEST"), "--denoise-variant", str(variant),
                            "--dump", prefix], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        assert "variant %d" % variant in r.stdout
        out[variant] = np.fromfile(prefix + ".denoisedCol.f32", np.float32)
    assert out[0].shape == (W * H * 3,) and np.isfinite(out[2]).all()
    scale = np.maximum(np.abs(out[0]), 1e-2)
    assert np.max(np.abs(out[2] - out[0]) / scale) < 1e-3
    r = subprocess.run([HEADLESS, "--scene", scene_path("cornell_crystal"), "--res", "96x54",
                        "--spp", "2", "--counter-seeds", "--denoise", "1", "--weights",
                        weights_path("BEST"), "--denoise-variant", "7"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "ptk_denoise" in r.stderr


def test_headless_rejects_bad_arguments(tmp_path):
    r = subprocess.run([HEADLESS], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "usage" in r.stderr
    r = subprocess.run([HEADLESS, "--scene", str(tmp_path / "missing.scene")], capture_output=True,
                       text=True, timeout=60)
    assert r.returncode == 1
    # sizes and counts the library refuses end in its message and status 1, not in an abort or
    # in host screens of that size
    for extra, what in ((["--res", "-5x10"], "resolution"), (["--res", "0x0"], "resolution"),
                        (["--res", "100000x100000"], "resolution"), (["--spp", "-3"], "spp"),
                        (["--depth", "0"], "max_depth"), (["--device", "99"], "device")):
        r = subprocess.run([HEADLESS, "--scene", scene_path("cornell")] + extra, capture_output=True,
                           text=True, timeout=60)
        assert r.returncode == 1 and what in r.stderr, (extra, r.returncode, r.stderr[-300:])
    r = subprocess.run([HEADLESS, "--scene", scene_path("cornell"), "--res", "16x9", "--spp", "0"],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "nan" not in r.stdout


def test_headless_training_epochs(tmp_path):
    """--train: the rendered frame becomes the target, each epoch renders a fresh noisy frame
    and takes one step; the weights written afterwards load back and differ from the start."""
    import ptk
    out_w = str(tmp_path / "trained.txt")
    cmd = [HEADLESS, "--scene", scene_path("cornell"), "--res", "96x54", "--spp", "64",
           "--counter-seeds", "--denoise", "1", "--weights", weights_path("box_grid"),
           "--train", "3", "--train-spp", "4", "--lr-exp", "4", "--save-weights", out_w,
           "--error-log", str(tmp_path / "ErrorLog.txt"), "--ppm", str(tmp_path / "t.ppm")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    rel = [float(l.split("relMSE")[1].split()[0]) for l in r.stdout.splitlines()
           if l.startswith("epoch")]
    assert len(rel) == 3 and all(np.isfinite(rel)) and all(x > 0 for x in rel)
    # the training log (DenoiserNN.cpp:1492-1514): header, one relMSE per epoch, total ms
    log = open(str(tmp_path / "ErrorLog.txt")).read().split("\n")
    assert log[0].startswith("Res: (96x54) Samples: 4 L Rate: ") and log[0].endswith(",")
    assert [float(l.rstrip(",")) for l in log[1:4]] == pytest.approx(rel, rel=1e-5)
    assert all(l.endswith(",") for l in log[1:4])
    assert float(log[4]) > 0 and log[5] == ""
    w0, w1 = ptk.load_weights(weights_path("box_grid")), ptk.load_weights(out_w)
    assert all(np.isfinite(b).all() for b in w1)
    assert any(not np.allclose(a, b, rtol=0, atol=1e-7) for a, b in zip(w0, w1))
    assert _read_ppm(str(tmp_path / "t.ppm")).shape == (54, 96, 3)


MULTIGPU = os.path.join(ROOT, "path-tracer_amd", "ptk_multigpu")


@pytest.mark.parametrize("denoise", [False, True])
def test_multi_gpu_driver_with_one_rank_writes_the_headless_image(tmp_path, denoise):
    """ptk_multigpu is the C++ host of the several-GPU frame: samples split across ranks, one
    exchange over RCCL (all-reduce of the colour planes, or ptk_pack_bands -> all-to-all ->
    ptk_merge_received -> band denoise -> in-place gather).  The test box has one GPU, so this
    runs the whole sequence with one rank -- communicator, collectives to itself, the merge
    kernel, the band denoise -- and the image must be the headless driver's, byte for byte
    (with one rank the merge adds nothing: the between-rank term is exactly zero)."""
    W, H, spp = 96, 54, 12
    base = ["--scene", scene_path("cornell_crystal"), "--res", "%dx%d" % (W, H), "--spp", str(spp)]
    extra = ["--denoise", "2", "--weights", weights_path("BEST")] if denoise else []
    one = str(tmp_path / "one.ppm")
    r = subprocess.run([HEADLESS] + base + ["--counter-seeds", "--auto-exposure", "--ppm", one] + extra,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    many = str(tmp_path / "many.ppm")
    r = subprocess.run([MULTIGPU, "--ranks", "1", "--rank", "0"] + base + ["--ppm", many] + extra,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Msamples/s" in r.stdout
    assert open(one).read() == open(many).read()
