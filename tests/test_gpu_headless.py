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


def test_headless_rejects_bad_arguments(tmp_path):
    r = subprocess.run([HEADLESS], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "usage" in r.stderr
    r = subprocess.run([HEADLESS, "--scene", str(tmp_path / "missing.scene")], capture_output=True,
                       text=True, timeout=60)
    assert r.returncode == 1


def test_headless_training_epochs(tmp_path):
    """--train: the rendered frame becomes the target, each epoch renders a fresh noisy frame
    and takes one step; the weights written afterwards load back and differ from the start."""
    import ptk
    out_w = str(tmp_path / "trained.txt")
    cmd = [HEADLESS, "--scene", scene_path("cornell"), "--res", "96x54", "--spp", "64",
           "--counter-seeds", "--denoise", "1", "--weights", weights_path("box_grid"),
           "--train", "3", "--train-spp", "4", "--lr-exp", "4", "--save-weights", out_w,
           "--ppm", str(tmp_path / "t.ppm")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    rel = [float(l.split("relMSE")[1].split()[0]) for l in r.stdout.splitlines()
           if l.startswith("epoch")]
    assert len(rel) == 3 and all(np.isfinite(rel)) and all(x > 0 for x in rel)
    w0, w1 = ptk.load_weights(weights_path("box_grid")), ptk.load_weights(out_w)
    assert all(np.isfinite(b).all() for b in w1)
    assert any(not np.allclose(a, b, rtol=0, atol=1e-7) for a, b in zip(w0, w1))
    assert _read_ppm(str(tmp_path / "t.ppm")).shape == (54, 96, 3)
