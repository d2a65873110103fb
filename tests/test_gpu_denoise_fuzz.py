"""Denoiser against the CPU restatement on random accumulators (scripts/gpu_denoise_fuzz.py):
images from 1x1 up, smaller than the 7x7 feature window and the filter window, around the tile,
strip and wave widths; filter half-widths 0..8; every weight file; exact zeros, flat planes and
large magnitudes.  Discrete feature planes bit for bit, the rest within 2e-5 / 5e-5 relative."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))


def test_random_accumulators_denoise_like_the_cpu_restatement(oracle):
    import gpu_denoise_fuzz
    bad = gpu_denoise_fuzz.fuzz(30, 5, verbose=False)
    assert not bad, "\n".join(bad)
