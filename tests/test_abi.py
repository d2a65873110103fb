"""The C-ABI library loads and exports every symbol include/ptk.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, "include", "ptk.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ptk_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_documented_entry_points():
    names = _declared()
    for must in ("ptk_create", "ptk_destroy", "ptk_set_constants", "ptk_render",
                 "ptk_read_accumulators", "ptk_auto_exposure", "ptk_denoise", "ptk_timing",
                 "ptk_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    import ptk
    assert os.path.exists(ptk.LIB_PATH), "libptk.so is not built (make -C path-tracer_amd)"
    lib = ctypes.CDLL(ptk.LIB_PATH)
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, "declared in ptk.h but not exported: %s" % missing
    assert sorted(ptk.EXPORTS) == _declared()


def test_constants_pod_is_3732_bytes():
    from oracle import oracle as O
    assert ctypes.sizeof(O.Constants) == 3732
    text = open(os.path.join(ROOT, "include", "ptk.h")).read()
    assert "#define PTK_CONSTANTS_BYTES 3732" in text


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import ptk
    monkeypatch.setattr(ptk, "_lib", None)
    monkeypatch.setattr(ptk, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ptk.PtkError):
        ptk.load()
