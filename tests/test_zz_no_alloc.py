"""ptk_auto_exposure_static -- the entry point that allocates nothing -- through the C ABI,
against ptk_auto_exposure on the same planes: the raw sum within 1e-9 relative (both add the
same fp32 per-pixel terms in double, in an order the atomics decide), the scaled exposure
within one fp32 ulp.  Two calls in a row, the second on a smaller image, share the library's
per-GPU array of partial sums; then the argument checks.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _planes_for(ctx, oracle, W, H, seed, spp):
    """Random non-negative colour sums written into a context's planes."""
    from conftest import scene_path
    consts = oracle.build_constants(oracle.load_scene(scene_path("cornell")), W, H, 4, 0.005, False)
    ctx.set_constants(consts)
    rng = np.random.default_rng(seed)
    planes = (rng.random((3, W * H), dtype=np.float32) * np.float32(spp)).astype(np.float32)
    ctx.write_planes(planes, spp)
    return planes


def test_static_exposure_matches_the_allocating_entry_point(oracle):
    import ptk
    import torch
    L = ptk.load()
    dev = torch.device("cuda", 0)
    out_e = torch.full((1,), -1.0, dtype=torch.float32, device=dev)
    out_raw = torch.full((1,), -1.0, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    ctx = ptk.Context(0)
    try:
        # the second image is smaller than the first and not a multiple of four pixels
        for (W, H, spp, seed) in ((640, 360, 12, 1), (149, 83, 5, 2)):
            planes = _planes_for(ctx, oracle, W, H, seed, spp)
            base, P = ctx.planes_device()
            assert P == W * H
            for formula in (ptk.EXPOSURE_CPU, ptk.EXPOSURE_SKEPU, ptk.EXPOSURE_MEAN):
                e_ref, raw_ref = ctx.auto_exposure(formula)
                ctx.synchronize()
                out_e.fill_(-1.0)
                out_raw.fill_(-1.0)
                torch.cuda.synchronize()
                rc = L.ptk_auto_exposure_static(C.c_void_p(base), P, W, H, spp, formula,
                                                C.c_void_p(out_e.data_ptr()),
                                                C.c_void_p(out_raw.data_ptr()), None)
                assert rc == 0, L.ptk_last_error().decode()
                torch.cuda.synchronize()
                e, raw = float(out_e.item()), float(out_raw.item())
                assert abs(raw - raw_ref) <= 1e-9 * abs(raw_ref)
                assert abs(e - e_ref) <= 1.2e-7 * abs(e_ref)
                # and against the CPU oracle on the same planes
                full = np.zeros((38, P), np.float32)
                full[0:3] = planes
                e_cpu, raw_cpu = oracle.auto_exposure(full, W, H, spp, formula)
                assert abs(raw - raw_cpu) <= 1e-9 * abs(raw_cpu)
                assert abs(e - e_cpu) <= 3e-4 * abs(e_cpu)   # the CPU side sums 230 k terms in fp32
            # one output pointer may be null
            out_raw.fill_(-1.0)
            torch.cuda.synchronize()
            rc = L.ptk_auto_exposure_static(C.c_void_p(base), P, W, H, spp, 0, None,
                                            C.c_void_p(out_raw.data_ptr()), None)
            assert rc == 0
            torch.cuda.synchronize()
            _, raw0 = oracle.auto_exposure(full, W, H, spp, 0)
            assert abs(float(out_raw.item()) - raw0) <= 1e-9 * abs(raw0)
        # planes at a 4-byte offset from a 16-byte boundary (a caller's sub-view), pixel count a
        # multiple of four: the 16-byte loads do not apply, the result is the same
        Wv, Hv, sppv = 64, 36, 3
        Pv = Wv * Hv
        host = (np.random.default_rng(5).random(3 * Pv, dtype=np.float32) * np.float32(sppv))
        backing = torch.zeros(3 * Pv + 1, dtype=torch.float32, device=dev)
        backing[1:] = torch.from_numpy(host).to(dev)
        view_ptr = backing.data_ptr() + 4
        assert view_ptr % 16 == 4
        torch.cuda.synchronize()
        rc = L.ptk_auto_exposure_static(C.c_void_p(view_ptr), Pv, Wv, Hv, sppv, 0,
                                        C.c_void_p(out_e.data_ptr()), C.c_void_p(out_raw.data_ptr()), None)
        assert rc == 0, L.ptk_last_error().decode()
        torch.cuda.synchronize()
        full = np.zeros((38, Pv), np.float32)
        full[0:3] = host.reshape(3, Pv)
        e_cpu, raw_cpu = oracle.auto_exposure(full, Wv, Hv, sppv, 0)
        assert abs(float(out_raw.item()) - raw_cpu) <= 1e-9 * abs(raw_cpu)
        assert abs(float(out_e.item()) - e_cpu) <= 3e-4 * abs(e_cpu)
        # argument checks: nothing is launched
        base, P = ctx.planes_device()
        bad = [
            (None, P, 149, 83, 5, 0, out_e.data_ptr(), out_raw.data_ptr()),
            (base, P, 149, 83, 5, 0, None, None),
            (base, P, 149, 83, 5, 3, out_e.data_ptr(), out_raw.data_ptr()),
            (base, (1 << 28) + 1, 149, 83, 5, 0, out_e.data_ptr(), out_raw.data_ptr()),
            (base, 0, 0, 0, 5, 0, out_e.data_ptr(), out_raw.data_ptr()),
            (base, P, 150, 83, 5, 0, out_e.data_ptr(), out_raw.data_ptr()),
            (base, P, 149, 83, 0, 0, out_e.data_ptr(), out_raw.data_ptr()),
        ]
        for a in bad:
            rc = L.ptk_auto_exposure_static(C.c_void_p(a[0]), a[1], a[2], a[3], a[4], a[5],
                                            C.c_void_p(a[6]), C.c_void_p(a[7]), None)
            assert rc == -1, a
            assert b"ptk_auto_exposure_static" in L.ptk_last_error()
    finally:
        ctx.close()
