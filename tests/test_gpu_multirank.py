"""The multi-rank path on the device: two ranks share the one GPU of the test box
and exchange through gloo (RCCL refuses two ranks on one device); the arithmetic
on the planes is the same code that runs over RCCL with one rank per GPU."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, scene_path, weights_path

pytestmark = pytest.mark.gpu

W, H, TOTAL_SPP, NWIN = 160, 90, 33, 2


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import multigpu
    import ptk
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    raw, _, _ = ptk.scene_constants(scene_path("MultiLightGlass"), W, H, 4, 0.005, True)
    w = ptk.load_weights(weights_path("BEST"))
    ctx = ptk.Context(0)
    ctx.set_constants(raw)
    ctx.set_stream_id(rank)
    n_local = multigpu.shard_spp(TOTAL_SPP, world, rank)
    ctx.render(n_local, ptk.SEED_COUNTER)
    planes = multigpu.device_planes(ctx, device)
    own = planes.clone()
    # (a) whole-frame merge (all-reduce), then the row-band denoiser
    n_total = multigpu.merge_planes(planes, n_local, ctx=ctx)
    torch.cuda.synchronize()
    assert ctx.sample_count == n_total
    den = multigpu.denoise_sharded(ctx, w, NWIN, device, n_total=n_total)
    torch.cuda.synchronize()
    merged, den_a = planes.cpu().numpy(), den.cpu().numpy()
    # (b) band-wise merge (all-to-all, halo rows included), then the same denoiser
    planes.copy_(own)
    torch.cuda.synchronize()
    ctx.sample_count = n_local
    n_b, lo, hi = multigpu.merge_to_bands(planes, n_local, W, H, halo=max(3, NWIN), ctx=ctx)
    den = multigpu.denoise_sharded(ctx, w, NWIN, device, n_total=n_b)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), planes=merged, denoised=den_a,
             n_total=n_total, banded=planes.cpu().numpy(), denoised_b=den.cpu().numpy(),
             span=np.array([lo, hi]), own=own.cpu().numpy(), n_local=n_local)
    ctx.close()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_match_a_single_context(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
    import multigpu
    import ptk
    world = 2
    port = 29600 + (os.getpid() % 300)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    assert int(r0["n_total"]) == TOTAL_SPP == int(r1["n_total"])
    assert np.array_equal(r0["planes"], r1["planes"], equal_nan=True)
    # after the gather both ranks hold the same full image
    assert np.array_equal(r0["denoised"], r1["denoised"], equal_nan=True)
    # The band-wise exchange (ptk_pack_bands, all-to-all, ptk_merge_received) gives each rank,
    # on its span, exactly what its stated rule gives in IEEE fp32 with the ranks in order ...
    own = [r0["own"].reshape(38, H, W), r1["own"].reshape(38, H, W)]
    n_r = [np.float32(int(r0["n_local"])), np.float32(int(r1["n_local"]))]
    n_all = np.float32(TOTAL_SPP)
    for r in (r0, r1):
        lo, hi = (int(x) for x in r["span"])
        x = [o[0:16, lo:hi] for o in own]
        v = [o[16:32, lo:hi] for o in own]
        sums = x[0] + x[1]
        mean = sums / n_all
        sdv = None
        for k in range(2):
            d = x[k] / n_r[k] - mean
            part = v[k] + n_r[k] * (d * d)
            sdv = part if sdv is None else sdv + part
        sd = np.stack([(sdv[3 * j] + sdv[3 * j + 1] + sdv[3 * j + 2]) / n_all for j in range(5)]
                      + [sdv[15] / n_all])
        a = r["banded"].reshape(38, H, W)[:, lo:hi]
        assert sums.dtype == np.float32 and sd.dtype == np.float32
        assert np.array_equal(a[0:16], sums, equal_nan=True)
        assert np.array_equal(a[16:32], sdv, equal_nan=True)
        assert np.array_equal(a[32:38], sd, equal_nan=True)
        # ... and leaves the rows outside the span as the rank rendered them
        k = 0 if r is r0 else 1
        rest = np.ones(H, bool)
        rest[lo:hi] = False
        assert np.array_equal(r["banded"].reshape(38, H, W)[:32, rest], own[k][:32, rest], equal_nan=True)
        # the whole-frame merge (all-reduce, tensor arithmetic with a reciprocal for the
        # scalar division) agrees: sums bit for bit, deviations to rounding
        b = r["planes"].reshape(38, H, W)[:, lo:hi]
        assert np.array_equal(a[0:16], b[0:16], equal_nan=True)
        scale = np.abs(b[16:]).max(axis=(1, 2), keepdims=True).clip(1e-6)
        assert (np.abs(a[16:] - b[16:]) / scale).max() < 1e-5
    # both ranks end with the same denoised frame (halo rows carry the same bits on both)
    assert np.array_equal(r0["denoised_b"], r1["denoised_b"], equal_nan=True)

    # the same job in one process: render each stream, merge by the same rule on the host
    raw, _, _ = ptk.scene_constants(scene_path("MultiLightGlass"), W, H, 4, 0.005, True)
    w = ptk.load_weights(weights_path("BEST"))
    ctx = ptk.Context(0)
    ctx.set_constants(raw)
    shards = []
    for rank in range(world):
        ctx.reset()
        ctx.set_stream_id(rank)
        n = multigpu.shard_spp(TOTAL_SPP, world, rank)
        ctx.render(n, ptk.SEED_COUNTER)
        shards.append((torch.from_numpy(ctx.read_planes().copy()), n))
    merged = torch.zeros_like(shards[0][0])
    merged[:16] = shards[0][0][:16] + shards[1][0][:16]
    mean_all = merged[:16] / float(TOTAL_SPP)
    for p, n in shards:
        merged[16:32] += p[16:32] + float(n) * (p[:16] / float(n) - mean_all) ** 2
    multigpu.refresh_std_dev(merged, TOTAL_SPP)
    got = torch.from_numpy(r0["planes"])
    assert torch.equal(got[:16], merged[:16])
    # the between-rank term differences nearly equal means (and the device divides by a
    # scalar through its reciprocal): compare against the scale of each plane
    scale = merged[16:].abs().amax(dim=1, keepdim=True).clamp_min(1e-6)
    assert torch.all(((got[16:] - merged[16:]).abs() / scale) < 1e-5)

    ctx.write_planes(r0["planes"], TOTAL_SPP)
    full = ctx.denoise(w, NWIN, 0)
    ref = full["denoised"].T
    # the same two-pass kernels on the same planes: bit-identical in both bands
    assert np.array_equal(r0["denoised"], ref, equal_nan=True)
    # and the band-merged frame: each rank's band denoised from its span of merged planes
    # equals the whole frame denoised from those planes stitched together
    stitched = r0["banded"].reshape(38, H, W).copy()
    lo1, hi1 = (int(x) for x in r1["span"])
    y0 = multigpu.row_band(H, world, 1)[0]
    stitched[:, y0:] = r1["banded"].reshape(38, H, W)[:, y0:]
    ctx.write_planes(stitched.reshape(38, H * W), TOTAL_SPP)
    full_b = ctx.denoise(w, NWIN, 0)
    ctx.close()
    assert np.array_equal(r0["denoised_b"], full_b["denoised"].T, equal_nan=True)
