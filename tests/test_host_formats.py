"""The on-disk formats either side of the hot path, through the product's own host facade
(libptk_host.so): .scene reader/writer (headers/Scene.h:127-274) and the weights
reader/writer (DenoiserNN.cpp:1603-1663).  No GPU work: only parsing and formatting."""
import glob
import os

import numpy as np
import pytest

from conftest import ROOT

SCENES = sorted(glob.glob(os.path.join(ROOT, "data", "scenes", "*.scene")))
WEIGHTS = sorted(glob.glob(os.path.join(ROOT, "data", "weights", "*.txt")))


@pytest.mark.parametrize("path", SCENES, ids=[os.path.basename(p) for p in SCENES])
def test_scene_survives_a_save_and_reload(path, tmp_path, oracle):
    import ptk
    out1, out2 = str(tmp_path / "a.scene"), str(tmp_path / "b.scene")
    ptk.resave_scene(path, out1)
    ptk.resave_scene(out1, out2)
    # data/scenes holds the reference's own Scenes/*.scene, which are outputs of its
    # Scene::SaveScene (headers/Scene.h:226-274): loading one and saving it again must
    # reproduce the file byte for byte
    assert open(out1, "rb").read() == open(path, "rb").read()
    assert open(out1).read() == open(out2).read()           # the writer is a fixed point
    a = ptk.scene_constants(path, 320, 180, 4, 0.005, True)
    b = ptk.scene_constants(out1, 320, 180, 4, 0.005, True)
    assert a[0] == b[0]                                      # identical 3732-byte POD
    assert a[1:] == b[1:]                                    # exposure, gamma
    # and the product's loader agrees with the test oracle's independent loader
    sc = oracle.load_scene(out1)
    assert bytes(oracle.build_constants(sc, 320, 180, 4, 0.005, True)) == a[0]


@pytest.mark.parametrize("path", WEIGHTS, ids=[os.path.basename(p) for p in WEIGHTS])
def test_weights_survive_a_save_and_reload(path, tmp_path, oracle):
    import ptk
    out1, out2 = str(tmp_path / "a.txt"), str(tmp_path / "b.txt")
    ptk.resave_weights(path, out1)
    ptk.resave_weights(out1, out2)
    # data/weights holds the reference's own Weights/*.txt, outputs of its OutputWeights
    # (DenoiserNN.cpp:1603-1624): a load and a save reproduce the file byte for byte
    assert open(out1, "rb").read() == open(path, "rb").read()
    w0, w1, w2 = ptk.load_weights(path), ptk.load_weights(out1), ptk.load_weights(out2)
    for x, y, z in zip(w0, w1, w2):
        # six significant digits are written (DenoiserNN.cpp:1603-1624)
        assert np.allclose(x, y, rtol=2e-6, atol=0)
        assert np.array_equal(y, z)
    assert len(open(out1).read().strip().split("\n")) == 3
    for x, y in zip(w0, oracle.load_weights(path)):
        assert np.array_equal(x, y)


def test_training_log_reproduces_the_reference_file(tmp_path):
    """tests/golden/ErrorLog.txt is the reference's own build/ErrorLog.txt (one training run:
    header, relMSE per epoch, total time; DenoiserNN.cpp:1492-1514).  Writing the same run
    through the facade's InitTraining header / AppendTrainingFile / EndTraining formatting
    gives the same bytes."""
    import ctypes as C
    import ptk
    ref = open(os.path.join(ROOT, "tests", "golden", "ErrorLog.txt")).read()
    lines = ref.split("\n")
    assert lines[0] == "Res: (480x270) Samples: 8 L Rate: 0.0001,"
    body = [l for l in lines[1:] if l]
    rel = np.float32([float(l.rstrip(",")) for l in body[:-1]])
    assert all(l.endswith(",") for l in body[:-1]) and not body[-1].endswith(",")
    H = ptk._load_host()
    H.ptkhost_write_training_log.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_float,
                                             C.POINTER(C.c_float), C.c_int, C.c_double]
    H.ptkhost_write_training_log.restype = C.c_int
    out = str(tmp_path / "ErrorLog.txt")
    rc = H.ptkhost_write_training_log(os.fsencode(out), 480, 270, 8, C.c_float(0.0001),
                                      rel.ctypes.data_as(C.POINTER(C.c_float)), len(rel),
                                      float(body[-1]))
    assert rc == 0
    assert open(out).read() == ref
    assert len(rel) == 53


def test_missing_files_are_reported(tmp_path):
    import ptk
    with pytest.raises(ptk.PtkError):
        ptk.resave_scene(str(tmp_path / "nope.scene"), str(tmp_path / "x"))
    with pytest.raises(ptk.PtkError):
        ptk.load_weights(str(tmp_path / "nope.txt"))


def test_scene_file_beyond_the_pod_keeps_every_shape(tmp_path):
    """A .scene file of 300 shapes, 70 materials and 14 important shapes: the POD the facade fills
    stops at the reference's limits, the tables it hands to ptk_set_scene hold everything, the
    hierarchy built for them verifies, and the file survives a save and reload."""
    import ptk
    rng = np.random.default_rng(8)
    lines = ["m %.3f/%.3f/%.3f 0 1 %d" % (*rng.uniform(0.2, 0.9, 3), 1 if k < 14 else 0)
             for k in range(70)]
    for k in range(300):
        c = rng.uniform(-8, 8, 3)
        if k % 2:
            lines += ["o 0", "d %.4f/%.4f/%.4f %.4f %d" % (c[0], c[1], c[2], rng.uniform(0.1, 0.5), k % 70)]
        else:
            h = rng.uniform(0.1, 0.6, 3)
            r = rng.uniform(0, 3, 3) if k % 4 == 0 else np.zeros(3)
            lines += ["o 1", "d %.4f/%.4f/%.4f %.4f/%.4f/%.4f %.4f/%.4f/%.4f %.3f/%.3f/%.3f %d" % (
                c[0], c[1], c[2], -h[0], -h[1], -h[2], h[0], h[1], h[2], r[0], r[1], r[2], k % 70)]
    lines += ["i %d" % k for k in range(14)]
    lines += ["c 0/0/-12 0/0/1 0/1/0 1/0/0 90 120", "pp 2.5 2.2"]
    path = str(tmp_path / "big.scene")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")

    raw, exposure, gamma = ptk.scene_constants(path, 64, 36)
    pod = ptk.scene_from_constants(raw)
    assert len(pod["shape_type"]) < 300 and len(pod["important_shapes"]) == 10
    scene, e2, g2 = ptk.scene_tables(path, 64, 36)
    assert (e2, g2) == (exposure, gamma) and abs(exposure - 2.5) < 1e-6
    assert len(scene["shape_type"]) == 300 and len(scene["mat_list"]) == 70
    assert list(scene["important_shapes"]) == list(range(14))
    n_pod = len(pod["shape_type"])
    assert np.array_equal(scene["shape_attrs"][:n_pod], pod["shape_attrs"])
    assert np.array_equal(scene["shape_mat"][:n_pod], pod["shape_mat"])
    stats = ptk.check_scene(scene)
    assert stats["leaves"] == 300

    again = str(tmp_path / "again.scene")
    ptk.resave_scene(path, again)
    scene2, _, _ = ptk.scene_tables(again, 64, 36)
    for key in ("shape_type", "shape_mat", "mat_list", "important_shapes"):
        assert np.array_equal(scene[key], scene2[key])
    assert np.allclose(scene["shape_attrs"], scene2["shape_attrs"], rtol=0, atol=2e-4)
