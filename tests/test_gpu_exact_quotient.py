"""Device self-check of the shared-denominator division (csrc/exact_ops.h): the reciprocal
against 1.0f / d for every significand at ten exponents, 4e9 quotients against a / d, and
zero / infinite / NaN numerators through the v_div_fixup form
(scripts/microbench/exact_ops_check.hip, built by the Makefile next to the library)."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

EXE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "path-tracer_amd",
                   "exact_ops_check")


def test_shared_denominator_division_is_ieee_division_on_the_device():
    assert os.path.exists(EXE), "exact_ops_check was not built (make -C path-tracer_amd)"
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr      # non-zero as soon as one result differs
    assert "differ" in r.stdout
