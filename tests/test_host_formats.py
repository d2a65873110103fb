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
    w0, w1, w2 = ptk.load_weights(path), ptk.load_weights(out1), ptk.load_weights(out2)
    for x, y, z in zip(w0, w1, w2):
        # six significant digits are written (DenoiserNN.cpp:1603-1624)
        assert np.allclose(x, y, rtol=2e-6, atol=0)
        assert np.array_equal(y, z)
    assert len(open(out1).read().strip().split("\n")) == 3
    for x, y in zip(w0, oracle.load_weights(path)):
        assert np.array_equal(x, y)


def test_missing_files_are_reported(tmp_path):
    import ptk
    with pytest.raises(ptk.PtkError):
        ptk.resave_scene(str(tmp_path / "nope.scene"), str(tmp_path / "x"))
    with pytest.raises(ptk.PtkError):
        ptk.load_weights(str(tmp_path / "nope.txt"))
