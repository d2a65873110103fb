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
             span=np.array([lo, hi]))
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
    # the band-wise exchange gives each rank the same merged values on its span, and the
    # same denoised frame
    for r in (r0, r1):
        lo, hi = (int(x) for x in r["span"])
        a = r["banded"].reshape(38, H, W)[:, lo:hi]
        b = r["planes"].reshape(38, H, W)[:, lo:hi]
        assert np.array_equal(a, b, equal_nan=True)
        assert np.array_equal(r["denoised_b"], r0["denoised"], equal_nan=True)

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
    ctx.close()
    ref = full["denoised"].T
    # the same two-pass kernels on the same planes: bit-identical in both bands
    assert np.array_equal(r0["denoised"], ref, equal_nan=True)
