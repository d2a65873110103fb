"""ptk_pack_bands -> all-to-all -> ptk_merge_received for any number of ranks, on one GPU: every
rank's planes are rendered in turn (its own sample stream), packed by the library, the
all-to-all is a reshuffle of the packed blocks done here, and each rank's merge is compared
with the stated rule evaluated in numpy fp32 with the ranks in order.  World sizes that do not
divide the height, more ranks than rows (empty bands), halos larger than a band.  The multi-GPU
bench runs this code with one rank per GPU over RCCL; the exchange itself is the caller's."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, scene_path

pytestmark = pytest.mark.gpu

CASES = [  # width, height, world, halo, total samples
    (40, 37, 3, 3, 10),
    (64, 24, 8, 3, 21),
    (33, 9, 13, 5, 30),      # more ranks than rows: bands of one row and empty bands
    (48, 64, 5, 8, 9),       # the halo of a filter wider than the feature window
    (32, 16, 16, 3, 16),     # one row and one sample per rank
    (20, 5, 8, 3, 17),
    (70, 41, 7, 20, 15),     # halo longer than any band
    (16, 33, 1, 3, 4),       # a single rank: the merge adds nothing
    (24, 128, 64, 3, 64),    # the largest world the merge kernel takes
]


@pytest.mark.parametrize("W,H,world,halo,total", CASES)
def test_band_exchange_for_any_world_size(W, H, world, halo, total):
    import torch
    sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
    import multigpu
    import ptk
    dev = torch.device("cuda", 0)
    raw, _, _ = ptk.scene_constants(scene_path("MultiLightGlass"), W, H, 4, 0.005, True)
    ctx = ptk.Context(0)
    ctx.set_constants(raw)
    spans = [ptk.band_layout(H, world, r, halo) for r in range(world)]
    rows_max = spans[0][2]
    assert all(s[2] == rows_max and s[1] - s[0] <= rows_max for s in spans)
    for r in range(world):          # the span is the band plus the halo, clipped to the frame
        y0, y1 = multigpu.row_band(H, world, r)
        assert spans[r][:2] == (max(0, y0 - halo), min(H, y1 + halo))
    block = 32 * rows_max * W
    own, n, send = [], [], []
    for r in range(world):
        ctx.reset()
        ctx.set_stream_id(r)
        n.append(multigpu.shard_spp(total, world, r))
        assert n[-1] >= 1
        ctx.render(n[-1], ptk.SEED_COUNTER)
        own.append(ctx.read_planes().copy())
        buf = torch.full((world * block,), float("nan"), device=dev)   # unwritten parts stay NaN
        torch.cuda.synchronize()        # the fill runs on torch's stream, the pack on the context's
        ctx.pack_bands(world, halo, buf.data_ptr(), buf.numel())
        torch.cuda.synchronize()
        send.append(buf)
        with pytest.raises(ptk.PtkError):
            ctx.pack_bands(world, halo, buf.data_ptr(), buf.numel() - 1)   # a buffer too small
    n_all = np.float32(total)
    for r in range(world):
        lo, hi, _ = spans[r]
        # what the all-to-all delivers to rank r: block r of every rank's send buffer
        recv = torch.cat([send[s][r * block:(r + 1) * block] for s in range(world)]).contiguous()
        torch.cuda.synchronize()
        # each block holds the sender's rows [lo, hi) of planes 0..31 and nothing else
        for s in range(world):
            blk = recv[s * block:(s + 1) * block].cpu().numpy().reshape(32, rows_max, W)
            want = own[s][:32].reshape(32, H, W)[:, lo:hi]
            assert np.array_equal(blk[:, :hi - lo], want, equal_nan=True)
            assert np.isnan(blk[:, hi - lo:]).all()
        ctx.write_planes(own[r], n[r])
        ctx.merge_received(world, r, halo, recv.data_ptr(), recv.numel(), n)
        assert ctx.sample_count == total
        got = ctx.read_planes().reshape(38, H, W)
        x = [o[0:16].reshape(16, H, W)[:, lo:hi] for o in own]
        v = [o[16:32].reshape(16, H, W)[:, lo:hi] for o in own]
        sums = x[0].copy()
        for k in range(1, world):
            sums = sums + x[k]
        mean = sums / n_all
        sdv = None
        for k in range(world):
            d = x[k] / np.float32(n[k]) - mean
            part = v[k] + np.float32(n[k]) * (d * d)
            sdv = part if sdv is None else sdv + part
        sd = np.stack([(sdv[3 * j] + sdv[3 * j + 1] + sdv[3 * j + 2]) / n_all for j in range(5)]
                      + [sdv[15] / n_all])
        assert sums.dtype == np.float32 and sd.dtype == np.float32
        assert np.array_equal(got[0:16, lo:hi], sums, equal_nan=True)
        assert np.array_equal(got[16:32, lo:hi], sdv, equal_nan=True)
        assert np.array_equal(got[32:38, lo:hi], sd, equal_nan=True)
        rest = np.ones(H, bool)
        rest[lo:hi] = False
        assert np.array_equal(got[:32, rest], own[r][:32].reshape(32, H, W)[:, rest], equal_nan=True)
    # arguments the layout cannot hold are refused
    buf = send[0]
    for bad_world, bad_rank in ((0, 0), (world, world), (world, -1), (65, 0)):
        with pytest.raises(ptk.PtkError):
            ctx.merge_received(bad_world, bad_rank, halo, buf.data_ptr(), buf.numel(), [1] * 65)
    ctx.close()


def test_any_row_band_of_any_frame_equals_the_whole_frame_result():
    """ptk_denoise_rows on random bands (one row, the first and last rows, the whole frame,
    empty) of random frames from 1x1 up, filter half-widths 0..8, both feature passes: the
    band's rows of the filtered colour and the weight sum equal the whole-frame call bit for
    bit and no other row is written."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
    import multigpu
    import ptk
    from conftest import weights_path
    rng = np.random.default_rng(77)
    raw_scene = scene_path("cornell_crystal")
    w = ptk.load_weights(weights_path("BEST"))
    ctx = ptk.Context(0)
    dev = torch.device("cuda", 0)
    fixed = [(1, 1), (1, 7), (7, 1), (62, 9), (63, 9), (64, 17), (125, 33)]
    for case in range(60):
        W, H = fixed[case] if case < len(fixed) else (int(rng.integers(1, 150)), int(rng.integers(1, 100)))
        P = W * H
        spp = int(rng.choice([1, 7, 64]))
        n = int(rng.integers(0, 9))
        variant = int(rng.choice([0, 2]))
        raw, _, _ = ptk.scene_constants(raw_scene, W, H, 4, 0.005, True)
        ctx.set_constants(raw)
        pl = np.zeros((38, P), np.float32)
        pl[0:16] = (rng.uniform(-1, 1, (16, P)) * spp).astype(np.float32)
        pl[32:38] = rng.uniform(1e-4, 0.5, (6, P))
        ctx.write_planes(pl, spp)
        full = ctx.denoise(w, n, variant)
        want = np.concatenate([full["denoised"].T, full["wcSum"].reshape(1, P)]).reshape(4, H, W)
        den, wc = multigpu.device_denoised(ctx, dev)
        y0 = int(rng.integers(0, H + 1))
        y1 = int(rng.integers(y0, H + 1))
        if case % 5 == 0:
            y0, y1 = 0, H
        elif case % 5 == 1 and H > 1:
            y0, y1 = H - 1, H
        den.fill_(-7.0)
        wc.fill_(-7.0)
        torch.cuda.synchronize()
        ctx.denoise_rows(w, n, variant, y0, y1)
        ctx.synchronize()
        got = np.concatenate([den.cpu().numpy().reshape(3, P), wc.cpu().numpy().reshape(1, P)]).reshape(4, H, W)
        what = "case %d: %dx%d N=%d variant %d rows [%d,%d)" % (case, W, H, n, variant, y0, y1)
        assert np.array_equal(got[:, y0:y1].view(np.uint32), want[:, y0:y1].view(np.uint32)), what
        outside = np.ones(H, bool)
        outside[y0:y1] = False
        assert (got[:, outside] == -7.0).all(), what
    ctx.close()
