"""The three-operation quotient by a correctly rounded reciprocal (make_recip / div_by in the
kernels) equals IEEE division: scripts/check_exact_quotient.c, every numerator significand for
a set of denominators."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_three_operation_quotient_is_the_ieee_quotient(tmp_path):
    exe = str(tmp_path / "check_exact_quotient")
    has_fma = "fma" in open("/proc/cpuinfo").read().split("flags", 1)[-1].split("\n", 1)[0].split()
    flags = ["-O2", "-fopenmp", "-ffp-contract=off"] + (["-mfma"] if has_fma else [])
    subprocess.run(["gcc"] + flags + ["-o", exe, os.path.join(ROOT, "scripts", "check_exact_quotient.c"), "-lm"],
                   check=True)
    r = subprocess.run([exe, "96" if has_fma else "8"], capture_output=True, text=True, timeout=600)
    sys.stdout.write(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 quotients differ, 0 first quotients" in r.stdout
