"""Samples-per-pixel sharding across the GPUs of one node.

One process per GPU.  Each rank renders its share of the samples of EVERY
pixel with its own counter-based seed stream (stream id = rank), then the
accumulator planes are summed across ranks with one all-reduce per frame
(RCCL over xGMI when the tensors live in HBM; gloo in the CPU tests).  The
denoiser is sharded by row bands: after the reduction every rank holds the
full planes, so a band needs no halo traffic; ranks denoise their own rows and
the bands are combined (zero-filled outside the band, one all-reduce).

The six running "stdDev" scalars are not sums over samples (the reference
updates them against the running mean, Renderers.cpp:860-887), so they cannot
be merged exactly.  The squared-deviation planes of the ranks are summed and
corrected with the between-rank term of the parallel-variance identity:

    sdv_total = sum_r sdv_r + sum_r n_r * (mean_r - mean)^2

which is exact for the Welford statistic and within O(ln n / n) relative of the
reference's order-dependent one (< 2 % for n >= 300, measured in
tests/test_multigpu.py).
"""
import torch
import torch.distributed as dist

N_SUM_PLANES = 16   # planes 0..15 are plain sums over samples
SDV0, N_SDV = 16, 16
SD0 = 32

# plane index of the sum each squared-deviation plane belongs to
_SUM_OF_SDV = list(range(0, 15)) + [15]


def all_reduce_(t, group=None, op=dist.ReduceOp.SUM):
    """In-place all-reduce.  RCCL reduces HBM tensors directly; under the gloo backend
    (CPU tests, single-GPU rehearsals of the multi-rank path) device tensors are staged
    through host memory."""
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


def shard_spp(total_spp, world_size, rank):
    """Samples rendered by `rank`: the first (total % world) ranks take one extra."""
    base, extra = divmod(int(total_spp), int(world_size))
    return base + (1 if rank < extra else 0)


def merge_planes(planes, n_local, group=None):
    """In-place cross-rank merge of a (38, P) plane tensor (CPU or GPU).

    planes: this rank's accumulators after rendering n_local samples.
    Returns the total sample count.  Afterwards every rank holds the merged
    sums, merged squared deviations and refreshed stdDev planes.
    """
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    n_t = torch.tensor([float(n_local)], dtype=torch.float64, device=planes.device)
    if world > 1:
        all_reduce_(n_t, group)
    n_total = int(round(float(n_t.item())))
    if world > 1:
        sums_local = planes[:N_SUM_PLANES].clone()
        all_reduce_(planes[:N_SUM_PLANES], group)
        if n_local > 0:
            mean_local = sums_local / float(n_local)
            mean_all = planes[:N_SUM_PLANES] / float(n_total)
            planes[SDV0:SDV0 + N_SDV] += float(n_local) * (mean_local - mean_all)[_SUM_OF_SDV] ** 2
        all_reduce_(planes[SDV0:SDV0 + N_SDV], group)
    refresh_std_dev(planes, n_total)
    return n_total


def refresh_std_dev(planes, n_total):
    """stdDev[k] = (sdv[3k] + sdv[3k+1] + sdv[3k+2]) / n, stdDev[5] = sdv[15] / n."""
    n = float(max(n_total, 1))
    for k in range(5):
        planes[SD0 + k] = (planes[SDV0 + 3 * k] + planes[SDV0 + 3 * k + 1]
                           + planes[SDV0 + 3 * k + 2]) / n
    planes[SD0 + 5] = planes[SDV0 + 15] / n


def row_band(height, world_size, rank):
    """Rows [y0, y1) denoised by `rank` (contiguous, sizes differ by at most one)."""
    base, extra = divmod(int(height), int(world_size))
    y0 = rank * base + min(rank, extra)
    return y0, y0 + base + (1 if rank < extra else 0)


class _DevicePlanes:
    """Exposes a context's HBM planes to torch through __cuda_array_interface__."""

    def __init__(self, ptr, n_planes, pixels):
        self.__cuda_array_interface__ = {
            "shape": (n_planes, pixels), "typestr": "<f4", "data": (ptr, False), "version": 2,
            "strides": None,
        }


def device_planes(ctx, device):
    """Zero-copy torch view (38, P) of the context's accumulator planes."""
    ptr, pixels = ctx.planes_device()
    return torch.as_tensor(_DevicePlanes(ptr, 38, pixels), device=device)


def device_denoised(ctx, device):
    """Zero-copy torch views of the denoiser outputs: (3, P) colour and (P,) weight sum."""
    d, w, _ = ctx.denoise_outputs_device()
    pixels = ctx.pixels
    return (torch.as_tensor(_DevicePlanes(d, 3, pixels), device=device),
            torch.as_tensor(_DevicePlanes(w, 1, pixels), device=device)[0])


def denoise_sharded(ctx, weights, n, device, variant=1, group=None):
    """Row-band sharded denoise: this rank filters its band, then the bands are combined
    with one all-reduce of the (zero-filled outside the band) colour planes.  Every rank
    must hold the merged accumulator planes (see merge_planes).  Returns the (3, P) view."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    y0, y1 = row_band(ctx.height, world, rank)
    den, wc = device_denoised(ctx, device)
    if world > 1:
        den.zero_()
        wc.zero_()
        # the context launches on its own stream: the clears must have landed first
        torch.cuda.current_stream(den.device).synchronize()
    ctx.denoise_rows(weights, n, variant, y0, y1)
    if world > 1:
        all_reduce_(den, group)
        all_reduce_(wc, group)
    return den
