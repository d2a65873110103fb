"""The register-window filter kernel reads neighbouring lanes through DPP operands written in
inline assembly, where the compiler cannot see two hardware hazards (a vector write followed
within two instructions by a DPP read of the same register; a transcendental result read by
the next instruction).  This compiles the kernel source to a gfx950 listing with the build's
flags and scans it (scripts/check_dpp_hazard.py): any offending pair fails the test."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_filter_kernel_listing_has_no_dpp_hazard(tmp_path):
    listing = tmp_path / "denoise.s"
    flags = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off",
             "-fno-fast-math", "-fno-slp-vectorize", "-I" + os.path.join(ROOT, "include")]
    subprocess.run([HIPCC] + flags + ["-S", "--cuda-device-only",
                                     os.path.join(ROOT, "path-tracer_amd", "csrc", "denoise.hip"),
                                     "-o", str(listing)], check=True, timeout=600)
    text = listing.read_text()
    assert text.count("_dpp") >= 300          # the kernel really is in the listing
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_dpp_hazard.py"),
                        str(listing)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 hazards" in r.stdout


def test_checker_flags_a_planted_hazard(tmp_path):
    bad = tmp_path / "bad.s"
    bad.write_text("\tv_add_f32_e32 v1, v2, v3\n"
                   "\tv_sub_f32_dpp v4, v1, v5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                   "\tv_exp_f32_e32 v7, v6\n"
                   "\tv_fmac_f32_dpp v8, v9, v7 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                   "\tv_add_f32_e32 v1, v2, v3\n\ts_nop 1\n"
                   "\tv_sub_f32_dpp v4, v1, v5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_dpp_hazard.py"),
                        str(bad)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1
    assert "2 hazards" in r.stdout, r.stdout


def _run_checker(path):
    return subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_dpp_hazard.py"),
                           str(path)], capture_output=True, text=True, timeout=60)


def test_checker_follows_branches_to_a_label(tmp_path):
    """A vector write just before a loop's back edge and a DPP read at the top of the loop are
    one wait state apart (the branch); the same with two s_nop after the label is fine."""
    dpp = "\tv_sub_f32_dpp v4, v1, v5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
    loop = ("\tv_mov_b32_e32 v9, 0\n\ts_nop 3\n.LBB0_1:\n%s" + dpp +
            "\ts_add_i32 s0, s0, 1\n\ts_cmp_lt_i32 s0, 8\n\tv_add_f32_e32 v1, v2, v3\n"
            "\ts_cbranch_scc1 .LBB0_1\n\ts_endpgm\n")
    bad = tmp_path / "loop_bad.s"
    bad.write_text(loop % "")
    r = _run_checker(bad)
    assert r.returncode == 1 and "1 hazards" in r.stdout, r.stdout
    good = tmp_path / "loop_good.s"
    good.write_text(loop % "\ts_nop 0\n")
    r = _run_checker(good)
    assert r.returncode == 0, r.stdout


def test_checker_flags_a_vector_write_of_exec(tmp_path):
    dpp = "\tv_sub_f32_dpp v4, v1, v5 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
    bad = tmp_path / "exec_bad.s"
    bad.write_text("\tv_cmpx_gt_f32_e32 v2, v3\n\ts_nop 2\n" + dpp)
    r = _run_checker(bad)
    assert r.returncode == 1 and "EXEC" in r.stdout, r.stdout
    good = tmp_path / "exec_good.s"
    good.write_text("\tv_cmpx_gt_f32_e32 v2, v3\n\ts_nop 4\n" + dpp)
    r = _run_checker(good)
    assert r.returncode == 0, r.stdout
