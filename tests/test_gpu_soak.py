"""scripts/gpu_soak.py in short: contexts created and destroyed, frame sizes from 1x1, the reference's
scenes and hierarchy scenes of 0 to 30 000 shapes, render / exposure / denoise (three variants) /
training step / post-process in random order.  Every call succeeds, the probe render has the same
bits afterwards, and the second half of the rounds leaves no device memory behind."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))


def test_random_call_sequences_leave_nothing_behind(monkeypatch):
    import gpu_soak
    monkeypatch.setattr(sys, "argv", ["gpu_soak.py", "120", "5"])
    assert gpu_soak.main() == 0
