"""Samples-per-pixel sharding across the GPUs of one node.

One process per GPU.  Each rank renders its share of the samples of EVERY
pixel with its own counter-based seed stream (stream id = rank), then the
accumulator planes are merged across ranks once per frame (RCCL over xGMI when
the tensors live in HBM; gloo in the CPU tests): an all-reduce when every rank
needs the whole frame (merge_planes; colour only when no denoise follows), or an
all-to-all that hands each rank the merged planes of its own row band plus the
halo rows its stencils read (merge_to_bands).  The denoiser is sharded by row
bands; the filtered bands are combined with one all-gather (denoise_sharded).

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
N_PLANES_TOTAL = 38

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


def merge_planes(planes, n_local, group=None, ctx=None):
    """In-place cross-rank merge of a (38, P) plane tensor (CPU or GPU).

    planes: this rank's accumulators after rendering n_local samples.
    Returns the total sample count.  Afterwards every rank holds the merged
    sums, merged squared deviations and refreshed stdDev planes.  When the
    tensor is a view of a context's planes, pass the context: its sample count
    becomes the merged total, which every later stage divides by.
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
    if ctx is not None:
        if planes.is_cuda:
            torch.cuda.current_stream(planes.device).synchronize()   # the context has its own stream
        ctx.sample_count = n_total
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


def band_with_halo(height, world_size, rank, halo):
    """Rows [lo, hi) a rank must hold to denoise its band: the band plus `halo` rows on
    either side (3 for the 7x7 feature statistics, N for a filter of half-width N > 3),
    clipped to the image."""
    y0, y1 = row_band(height, world_size, rank)
    return max(0, y0 - halo), min(int(height), y1 + halo)


def _all_to_all_single(recv, send, group=None):
    if send.is_cuda and dist.get_backend(group) == "gloo":
        h_send, h_recv = send.cpu(), torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(h_recv, h_send, group=group)
        recv.copy_(h_recv)
    else:
        dist.all_to_all_single(recv, send, group=group)


def _all_gather_into(out, part, group=None):
    """out: (world, *part.shape).  The collective sees it as the concatenation of the parts
    along their first dimension, the one form every backend accepts."""
    flat = out.view(out.shape[0] * part.shape[0], *part.shape[1:])
    if part.is_cuda and dist.get_backend(group) == "gloo":
        h_out = torch.empty(flat.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h_out, part.cpu(), group=group)
        flat.copy_(h_out)
    else:
        dist.all_gather_into_tensor(flat, part, group=group)


def merge_to_bands(planes, n_local, width, height, halo=3, group=None, ctx=None):
    """Cross-rank merge that leaves each rank with the merged accumulators of ITS ROW BAND
    plus `halo` rows on either side -- what the row-band denoiser reads -- instead of the
    whole frame: one all-to-all in which rank s sends rank d the rows of d's band (halo
    included) of its 16 sum and 16 squared-deviation planes.  Every GPU talks to every
    other directly, so the exchange uses all point-to-point links at once and each rank
    receives (world - 1) / world of one frame's planes, half of what the ring all-reduce
    of merge_planes moves; the halo rows ride in the same exchange.

    The merge rule is merge_planes': sums add, squared deviations add with the
    between-rank term n_r (mean_r - mean)^2, stdDev = sdv / n.  Ranks are added in
    rank order, so a row two ranks both hold (a halo row) gets the same bits on both.
    planes: (38, P) tensor; rows outside [lo, hi) are left as they were and must not be
    read afterwards.  Returns (n_total, lo, hi).

    When `planes` is the zero-copy view of a context's HBM planes (pass the context), the
    work on either side of the all-to-all is the library's own: ptk_pack_bands lays the send
    buffer out with one strided device copy per destination and ptk_merge_received merges
    the receive buffer in one kernel that reads it once (IEEE fp32, ranks in order).  The
    tensor code below states the same rule for host tensors (the CPU tests)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    W, H = int(width), int(height)
    lo, hi = band_with_halo(H, world, rank, halo)
    if world == 1:
        refresh_std_dev(planes, n_local)
        if ctx is not None:
            ctx.sample_count = int(n_local)
        return int(n_local), lo, hi
    spans = [band_with_halo(H, world, d, halo) for d in range(world)]
    rows_max = max(b - a for a, b in spans)
    dev = planes.device
    counts = torch.zeros(world, dtype=torch.float64, device=dev)
    counts[rank] = float(n_local)
    all_reduce_(counts, group)
    n_r = [int(round(float(x))) for x in counts.tolist()]
    n_total = sum(n_r)

    if ctx is not None and planes.is_cuda:
        send = torch.empty((world, 32, rows_max, W), dtype=planes.dtype, device=dev)
        torch.cuda.current_stream(dev).synchronize()      # the context has its own stream
        ctx.pack_bands(world, halo, send.data_ptr(), send.numel())
        recv = torch.empty_like(send)
        _all_to_all_single(recv, send, group)
        del send
        torch.cuda.current_stream(dev).synchronize()
        ctx.merge_received(world, rank, halo, recv.data_ptr(), recv.numel(), n_r)
        return n_total, lo, hi

    img = planes.view(N_PLANES_TOTAL, H, W)
    send = torch.zeros((world, 32, rows_max, W), dtype=planes.dtype, device=dev)
    for d, (a, b) in enumerate(spans):
        send[d, :, :b - a] = img[0:32, a:b]
    recv = torch.empty_like(send)
    _all_to_all_single(recv, send, group)
    del send
    rows = hi - lo
    got = recv[:, :, :rows]                        # (world, 32, rows, W): rank s's slice of my span
    sums = got[0, 0:16].clone()
    for s_ in range(1, world):
        sums += got[s_, 0:16]
    mean_all = sums / float(max(n_total, 1))
    sdv = torch.zeros_like(sums)
    for s_ in range(world):
        part = got[s_, 16:32].clone()
        if n_r[s_] > 0:
            part += float(n_r[s_]) * (got[s_, 0:16] / float(n_r[s_]) - mean_all)[_SUM_OF_SDV] ** 2
        sdv = part if s_ == 0 else sdv + part
    img[0:16, lo:hi] = sums
    img[SDV0:SDV0 + N_SDV, lo:hi] = sdv
    n = float(max(n_total, 1))
    for k in range(5):
        img[SD0 + k, lo:hi] = (sdv[3 * k] + sdv[3 * k + 1] + sdv[3 * k + 2]) / n
    img[SD0 + 5, lo:hi] = sdv[15] / n
    if ctx is not None:
        if planes.is_cuda:
            torch.cuda.current_stream(dev).synchronize()   # the context has its own stream
        ctx.sample_count = n_total
    return n_total, lo, hi


def exchange_bytes(width, height, world_size, halo=3, features=True):
    """Bytes one rank RECEIVES in merge_to_bands (features) or in the ring all-reduce of the
    three colour planes (no features), for reporting next to a timing."""
    if world_size <= 1:
        return 0
    if not features:
        return int(2 * (world_size - 1) / world_size * 3 * width * height * 4)
    lo, hi = band_with_halo(height, world_size, world_size // 2, halo)
    return int((world_size - 1) * 32 * (hi - lo) * width * 4)


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


def denoise_sharded(ctx, weights, n, device, variant=0, group=None, n_total=None):
    """Row-band sharded denoise: this rank filters its band, then the bands are gathered
    with one all-gather of the band slices (12 bytes per pixel of the frame in all, against
    eight times that for an all-reduce of zero-filled frames).  Every rank must hold the
    merged accumulators of its band and max(3, n) rows either side (merge_to_bands, or the
    whole frame after merge_planes) and the merged sample count (checked when n_total is
    given).  variant 0 is the two-pass denoiser every other caller uses; all ranks must pass
    the same one.  Returns the (3, P) view of the whole denoised frame, valid on every rank."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if n_total is not None and ctx.sample_count != n_total:
        raise ValueError("denoise_sharded: the context counts %d samples, the merged planes hold %d"
                         % (ctx.sample_count, n_total))
    H, W = ctx.height, ctx.width
    y0, y1 = row_band(H, world, rank)
    den, wc = device_denoised(ctx, device)
    ctx.denoise_rows(weights, n, variant, y0, y1)       # returns when the band is written
    if dist.is_initialized():        # with one rank too: the same collective, gathering from itself
        gather_bands_(den.view(3, H, W), wc.view(H, W), world, rank, group)
        torch.cuda.current_stream(den.device).synchronize() if den.is_cuda else None
    return den


def gather_bands_(den, wc, world, rank, group=None):
    """Every rank's band of the three colour planes den (3, H, W) and of the weight-sum plane
    wc (H, W) to every rank, in place.  A band's rows of one plane are contiguous and the
    bands tile the plane in rank order, so when the height divides evenly each plane IS the
    concatenation an all-gather produces and the collective writes the peers' bands
    straight to where they belong (four all-gathers, no staging).  Ragged bands go through
    one padded all-gather."""
    H, W = wc.shape
    if H % world == 0 and not (den.is_cuda and dist.get_backend(group) == "gloo"):
        rows = H // world
        for plane in (den[0], den[1], den[2], wc):
            dist.all_gather_into_tensor(plane, plane[rank * rows:(rank + 1) * rows], group=group)
        return
    y0, y1 = row_band(H, world, rank)
    rows_max = max(b - a for a, b in (row_band(H, world, r) for r in range(world)))
    part = torch.empty((4, rows_max, W), dtype=den.dtype, device=den.device)
    part[0:3, :y1 - y0] = den[:, y0:y1]
    part[3, :y1 - y0] = wc[y0:y1]
    full = torch.empty((world, 4, rows_max, W), dtype=den.dtype, device=den.device)
    _all_gather_into(full, part, group)
    for r in range(world):
        a, b = row_band(H, world, r)
        if r != rank:
            den[:, a:b] = full[r, 0:3, :b - a]
            wc[a:b] = full[r, 3, :b - a]
