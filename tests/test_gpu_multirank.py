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
    n_total = multigpu.merge_planes(planes, n_local)
    torch.cuda.synchronize()
    ctx.sample_count = n_total
    den = multigpu.denoise_sharded(ctx, w, NWIN, device, variant=rank % 2)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), planes=planes.cpu().numpy(),
             denoised=den.cpu().numpy(), n_total=n_total)
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
    # bands were filtered by different variants (two-pass on rank 0, fused on rank 1);
    # after the exchange both ranks hold the same full image
    assert np.array_equal(r0["denoised"], r1["denoised"], equal_nan=True)

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
    y0, y1 = multigpu.row_band(H, world, 0)
    band0 = np.zeros(H * W, bool)
    band0[y0 * W:y1 * W] = True
    # rank 0's band: same two-pass kernels, bit-identical; rank 1's band: the fused kernel
    assert np.array_equal(r0["denoised"][:, band0], ref[:, band0], equal_nan=True)
    a, b = r0["denoised"][:, ~band0].astype(np.float64), ref[:, ~band0].astype(np.float64)
    assert np.all(np.abs(a - b) <= 2e-5 * np.abs(b) + 1e-7)
