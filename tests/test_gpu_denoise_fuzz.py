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


@pytest.mark.parametrize("variant", [0, 2])
def test_frames_with_nan_and_infinities_denoise_like_the_cpu_restatement(oracle, variant):
    """NaN and infinities sprinkled into the accumulators (a path that divided by a zero pdf):
    the reference-arithmetic pass equals the restatement as everywhere else, NaN where it has
    NaN; the tolerance-held pass may differ only where the restatement's smooth statistics are
    not finite, and equals it in bandwidths, filtered colour and weight sums."""
    import gpu_denoise_fuzz
    bad = gpu_denoise_fuzz.fuzz(24, 9, verbose=False, variant=variant, kinds=("nonfinite",))
    assert not bad, "\n".join(bad)


def test_post_process_on_random_accumulators(oracle):
    """scripts/gpu_post_process_fuzz.py: display metrics 0..6 with random exposure and gamma on
    images of every size from 1x1 up and on values a rendered frame does not reach (negative
    sums, which g = 1 passes through and any other gamma turns into NaN; exact zeros; subnormal,
    tiny and huge values; NaN and infinities): float screen within 2e-6 of the restatement
    (2e-5 beyond [1e-3, 1e3]), same class where it is not finite, 8-bit image = the quantised
    float screen."""
    import gpu_post_process_fuzz
    bad = gpu_post_process_fuzz.fuzz(40, 3, verbose=False)
    assert not bad, "\n".join(bad)


def test_exposure_reduce_on_random_colour_planes(oracle):
    """scripts/gpu_exposure_fuzz.py: the three exposure formulas on images from 1x1 to beyond one
    round of workgroups, pixel counts that are not multiples of four, negative, cancelling, huge,
    subnormal and non-finite values: the raw sum within 1e-9 of the restatement's double sum,
    the same class where that is not finite."""
    import gpu_exposure_fuzz
    bad = gpu_exposure_fuzz.fuzz(40, 3, verbose=False)
    assert not bad, "\n".join(bad)
