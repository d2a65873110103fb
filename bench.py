#!/usr/bin/env python3
"""Benchmark of the path-trace hot path on MI355X.

A step is one ptk_render call of --spp-per-step samples per pixel on
cornell_crystal at 1920x1080 (BASELINE.json configs[1]: 2000 spp, render only;
the default 10 steps x 200 spp are exactly that job).  The accumulators live in
HBM for the whole run and seeds are generated on the device, so nothing crosses
PCIe inside the timed region.  With N > 1 ranks every rank renders the same
number of samples of every pixel on its own seed stream (weak scaling: the
samples of the frame are split across the GPUs) and the frame ends, inside the
timed region, with ONE accumulation exchange: the all-reduce of the colour planes
over RCCL.  After the timed region the multi-rank run also times the two
eight-GPU jobs of BASELINE.json at 3840x2160 (objects "config4" and "config5" of
the result line): the sample-split render with its all-reduce, and the feature
render + band-wise merge + row-band denoiser + gather.

Prints one JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import multigpu  # noqa: E402
import ptk  # noqa: E402

PEAK_FP32_TFLOPS = 157.3   # MI355X vector fp32 peak (FMA = 2 flop)
PEAK_HBM_GBS = 8000.0      # MI355X HBM3E peak
PEAK_VALU_ISSUE_T = 78.6   # lane-instructions/s: 256 CUs x 4 SIMDs x 32 lanes/cycle x 2.4 GHz
PROFILE_SUMMARY = os.path.join(ROOT, "profiles", "r03_profile_summary.json")
PROFILE_STALE_TOLERANCE = 0.02   # a committed profile speaks for this run only if its launch time agrees


def load_opcount(scene):
    """Algorithmic operations per sample, counted by the op-counting build of the CPU
    restatement (profiles/opcount.json, written by scripts/count_ops.py)."""
    path = os.path.join(ROOT, "profiles", "opcount.json")
    with open(path) as f:
        table = json.load(f)
    return table[scene]


def load_profile(samples_per_launch, features):
    """Per-launch PMC figures of the render kernel from the committed rocprofv3 passes
    (scripts/collect_profiles.sh), valid only for the launch size they were taken at."""
    try:
        with open(PROFILE_SUMMARY) as f:
            doc = json.load(f)
        if doc["bench"]["config"]["samples_per_step_per_gpu"] != samples_per_launch:
            return None
        want = "render_kernel<4, %s," % ("true" if features else "false")
        k = next(v for name, v in doc["kernels"].items() if name.startswith(want))
        return k
    except (OSError, KeyError, StopIteration, ValueError):
        return None


def cpu_baseline(scene, W, H, depth):
    """The CPU oracle timed on the host cores: a bounded sample of the same workload."""
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    threads = max(1, min(threads, 16))
    sc = O.load_scene(os.path.join(ROOT, "data", "scenes", scene + ".scene"))
    consts = O.build_constants(sc, W, H, depth, 0.005, False)
    acc = O.new_accumulators(W, H)
    seed = (0x9E3779B97F4A7C15, 0xD1B54A32D192ED03)
    O.render(consts, acc, 1, 0, seed, threads=threads)      # warm (page in, spawn the team)
    spp = 4
    t0 = time.perf_counter()
    O.render(consts, acc, spp, 1, seed, threads=threads)
    dt = time.perf_counter() - t0
    # keep the whole leg near 10-30 s: extend once if the first sample was short
    if dt < 5.0:
        more = int(min(32, max(4, spp * 10.0 / max(dt, 1e-3))))
        t0 = time.perf_counter()
        O.render(consts, acc, more, 1 + spp, seed, threads=threads)
        dt = time.perf_counter() - t0
        spp = more
    out = {
        "value": W * H * spp / dt / 1e6, "unit": "Msamples/s", "cores": threads, "kind": "port",
        "sample": "%s %dx%d, %d spp, maxDepth %d, OpenMP over rows with pre-generated "
                  "reference seeds (oracle/ptk_oracle.c)" % (scene, W, H, spp, depth),
    }
    # one thread on a quarter-size image (the reference's own default is sequential)
    w4, h4 = W // 2, H // 2
    c4 = O.build_constants(sc, w4, h4, depth, 0.005, False)
    a4 = O.new_accumulators(w4, h4)
    t0 = time.perf_counter()
    O.render(c4, a4, 2, 0, seed, threads=1)
    out["single_thread_msamples_per_s"] = w4 * h4 * 2 / (time.perf_counter() - t0) / 1e6
    # denoiser inference on the host cores: configs[2]'s scene at 8 spp, N = 1
    scd = O.load_scene(os.path.join(ROOT, "data", "scenes", "MultiLightGlass.scene"))
    cd = O.build_constants(scd, W, H, depth, 0.005, True)
    ad = O.new_accumulators(W, H)
    O.render(cd, ad, 8, 0, seed, threads=threads)
    wts = O.load_weights(os.path.join(ROOT, "data", "weights", "BEST.txt"))
    t0 = time.perf_counter()
    _, var = O.forward_prop(ad, W, H, 8, wts, threads=threads)
    t1 = time.perf_counter()
    O.filter(ad, var, W, H, 8, 1, threads=threads)
    t2 = time.perf_counter()
    out["denoise_ms"] = (t2 - t0) * 1e3
    out["denoise_features_ms"] = (t1 - t0) * 1e3
    out["denoise_filter_ms"] = (t2 - t1) * 1e3
    out["denoise_sample"] = "MultiLightGlass %dx%d, 8 spp, BEST.txt, N=1, %d threads" % (W, H, threads)
    return out


def denoise_probe(ctx, W, H, depth):
    """BASELINE.json configs[2]: MultiLightGlass 1080p, 128 spp, BEST.txt, N = 1."""
    raw, _, _ = ptk.scene_constants(os.path.join(ROOT, "data", "scenes", "MultiLightGlass.scene"),
                                    W, H, depth, 0.005, True)
    w = ptk.load_weights(os.path.join(ROOT, "data", "weights", "BEST.txt"))
    ctx.set_constants(raw)
    ctx.render(128, ptk.SEED_COUNTER)
    render_ms = ctx.timing().render_ms
    out = {"workload": "MultiLightGlass %dx%d 128 spp + denoise N=1 BEST.txt" % (W, H),
           "render_msamples_per_s": W * H * 128 / render_ms / 1e3}
    def timed_denoise(variant):
        ts = []
        for _ in range(12):
            t0 = time.perf_counter()
            ctx.denoise(w, 1, variant, want_outputs=False)
            wall = (time.perf_counter() - t0) * 1e3
            t = ctx.timing()
            ts.append((t.denoise_ms, t.features_ms, t.filter_ms, wall))
        ts = ts[2:]
        return [sum(x[i] for x in ts) / len(ts) for i in range(4)]

    avg = timed_denoise(0)
    out["two_pass_ms"] = avg[0]
    # host clock around the whole ptk_denoise call (argument checks, both launches, one wait)
    out["call_wall_ms"] = avg[3]
    out["features_ms"], out["filter_ms"] = avg[1], avg[2]
    # algorithmic bytes per pixel: features 72 read + 28 written, filter 116 read + 16 written
    out["features_gbs"] = W * H * 100 / avg[1] / 1e6
    out["filter_gbs"] = W * H * 132 / avg[2] / 1e6
    # variant 2: the feature pass in its tolerance-held form (smooth statistics formed a
    # cheaper way, discrete ones bit for bit; tests/test_gpu_denoise_fast.py holds it to 2e-5)
    fast = timed_denoise(2)
    out["tolerance_held_features"] = {
        "variant": 2, "two_pass_ms": fast[0], "features_ms": fast[1], "filter_ms": fast[2],
        "features_gbs": W * H * 100 / fast[1] / 1e6,
        "features_frac": W * H * 100 / fast[1] / 1e6 / PEAK_HBM_GBS}
    es = []
    for _ in range(12):
        ctx.auto_exposure(ptk.EXPOSURE_CPU)
        es.append(ctx.timing().exposure_ms)
    e = sum(es[2:]) / len(es[2:])
    out["exposure_ms"] = e
    out["exposure_gbs"] = W * H * 12 / e / 1e6
    ps = []
    for _ in range(12):
        ctx.post_process(0, 1.0, 2.2, want_post=False)
        ps.append(ctx.timing().post_ms)
    p = sum(ps[2:]) / len(ps[2:])
    out["post_process_ms"] = p
    out["post_process_gbs"] = W * H * 24 / p / 1e6
    # one training step's reductions (relMSE + back-prop sums) with this frame as its own target
    ctx.save_target()
    ts = []
    for _ in range(6):
        ctx.train_step(w, 1, 1e-6)
        ts.append(ctx.timing().train_ms)
    out["train_reduce_ms"] = sum(ts[1:]) / len(ts[1:])
    # what a plain device copy achieves on this board, for scale (read + write counted)
    n = 1 << 28
    src = torch.empty(n, dtype=torch.float32, device="cuda")
    dst = torch.empty_like(src)
    dst.copy_(src)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(5):
        dst.copy_(src)
    ev1.record()
    torch.cuda.synchronize()
    out["device_copy_gbs"] = 5 * 2 * n * 4 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9
    # roofline fractions of the two stencils: against the 8 TB/s HBM peak and against what a
    # plain copy reaches on this board
    for k in ("features", "filter", "exposure", "post_process"):
        out[k + "_frac"] = out[k + "_gbs"] / PEAK_HBM_GBS
        out[k + "_frac_of_copy"] = out[k + "_gbs"] / out["device_copy_gbs"]
    # The event pair around a launch of a few microseconds also times the gaps on either side.
    # The committed kernel trace has the kernels' own durations (1920x1080 only): quoted here,
    # marked as such, with the rates they imply.
    if (W, H) == (1920, 1080):
        try:
            with open(PROFILE_SUMMARY) as f:
                ks = json.load(f)["kernels"]
            per_px = {"features_mlp_kernel<false>": 100, "features_mlp_fast_kernel<false>": 100,
                      "filter_rows_kernel": 132, "exposure_kernel<0>": 12, "post_process_kernel<4>": 24}
            out["rocprof_kernel_trace"] = {
                name: {"avg_us": ks[name]["avg_ms"] * 1e3, "min_us": ks[name]["min_ms"] * 1e3,
                       "gbs_at_avg": W * H * b / ks[name]["avg_ms"] / 1e6}
                for name, b in per_px.items() if name in ks}
            out["rocprof_kernel_trace"]["source"] = "profiles/%s (committed run)" % os.path.basename(PROFILE_SUMMARY)
        except (OSError, KeyError, ValueError):
            pass
    return out


def denoise_4k_probe(ctx, depth):
    """The denoiser at 3840x2160 on a rendered frame (MultiLightGlass, 32 spp with features,
    BEST.txt, N = 1).  The 1080p frame's 240 MB of filter input fit the 256 MB Infinity
    Cache, and the feature pass that runs just before the filter has read most of them; at
    this size (1.26 GB of planes) the stencils stream from HBM, so these are the figures to
    hold against the HBM peak."""
    W, H = 3840, 2160
    raw, _, _ = ptk.scene_constants(os.path.join(ROOT, "data", "scenes", "MultiLightGlass.scene"),
                                    W, H, depth, 0.005, True)
    w = ptk.load_weights(os.path.join(ROOT, "data", "weights", "BEST.txt"))
    ctx.set_constants(raw)
    ctx.render(32, ptk.SEED_COUNTER)
    out = {"workload": "MultiLightGlass %dx%d 32 spp + denoise N=1 BEST.txt" % (W, H),
           "render_msamples_per_s": W * H * 32 / ctx.timing().render_ms / 1e3}
    for variant, tag in ((0, ""), (2, "_tolerance_held")):
        ts = []
        for _ in range(10):
            ctx.denoise(w, 1, variant, want_outputs=False)
            t = ctx.timing()
            ts.append((t.features_ms, t.filter_ms))
        ts = ts[2:]
        f_ms, g_ms = (sum(x[i] for x in ts) / len(ts) for i in range(2))
        out["features%s_ms" % tag] = f_ms
        out["features%s_gbs" % tag] = W * H * 100 / f_ms / 1e6
        out["features%s_frac" % tag] = W * H * 100 / f_ms / 1e6 / PEAK_HBM_GBS
        if variant == 0:
            out["filter_ms"] = g_ms
            out["filter_gbs"] = W * H * 132 / g_ms / 1e6
            out["filter_frac"] = out["filter_gbs"] / PEAK_HBM_GBS
    es = []
    for _ in range(10):
        ctx.auto_exposure(ptk.EXPOSURE_CPU)
        es.append(ctx.timing().exposure_ms)
    e = sum(es[2:]) / len(es[2:])
    out["exposure_ms"] = e
    out["exposure_gbs"] = W * H * 12 / e / 1e6
    out["exposure_frac"] = out["exposure_gbs"] / PEAK_HBM_GBS
    return out


def hierarchy_probe(ctx, W, H):
    """Scenes beyond the reference's 50-shape POD (ptk_set_scene): a synthetic grid of ~97 k
    spheres and boxes behind the hierarchy, 8 spp, no features."""
    import scenes
    sc = scenes.grid_scene(46, W, H)
    st = ctx.set_scene(sc)
    ctx.render(2, ptk.SEED_COUNTER)
    ctx.render(8, ptk.SEED_COUNTER)
    t = ctx.timing()
    return {"workload": "synthetic grid, %d shapes, %dx%d, 8 spp, maxDepth 4" % (st["num_shapes"], W, H),
            "msamples_per_s": t.render_samples / t.render_ms / 1e3,
            "mrays_per_s": t.render_rays / t.render_ms / 1e3,
            "nodes": st["nodes"], "depth": st["depth"], "host_build_ms": st["build_ms"],
            "device_mb": st["device_bytes"] / 1e6}


def _barrier(args, gpu):
    torch.cuda.synchronize()
    dist.barrier(device_ids=[gpu]) if args.backend == "nccl" else dist.barrier()
    torch.cuda.synchronize()


def _max_over_ranks(seconds, device):
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    multigpu.all_reduce_(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def config2_split_leg(ctx, args, rank, world, gpu, device):
    """BASELINE.json configs[1] as one job on all ranks (strong scaling): cornell_crystal
    1920x1080, the frame's 2000 samples per pixel split across the ranks, one all-reduce of
    the colour planes."""
    W, H = 1920, 1080
    total = 2000
    raw, _, _ = ptk.scene_constants(os.path.join(ROOT, "data", "scenes", "cornell_crystal.scene"),
                                    W, H, args.depth, 0.005, False)
    ctx.set_constants(raw)
    ctx.set_stream_id(rank)
    planes = multigpu.device_planes(ctx, device)
    n_local = multigpu.shard_spp(total, world, rank)
    ctx.render(8, ptk.SEED_COUNTER)
    multigpu.all_reduce_(planes[0:3])
    ctx.reset()
    ctx.synchronize()
    _barrier(args, gpu)
    t0 = time.perf_counter()
    ctx.render(n_local, ptk.SEED_COUNTER)
    t1 = time.perf_counter()
    multigpu.all_reduce_(planes[0:3])
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    _barrier(args, gpu)
    elapsed = _max_over_ranks(time.perf_counter() - t0, device)
    return {"workload": "Scenes/cornell_crystal.scene %dx%d, %d spp split over %d GPUs (strong scaling), "
                        "all-reduce of the colour planes" % (W, H, total, world),
            "msamples_per_s": float(W) * H * total / elapsed / 1e6, "frame_ms": elapsed * 1e3,
            "render_ms_rank0": (t1 - t0) * 1e3, "exchange_ms_rank0": (t2 - t1) * 1e3,
            "exchange_bytes_per_rank": multigpu.exchange_bytes(W, H, world, features=False)}


def config4_leg(ctx, args, rank, world, gpu, device):
    """BASELINE.json configs[3]: cornell_crystal 3840x2160, the frame's samples split across
    the ranks (625 per GPU: 5000 spp on eight), one all-reduce of the colour planes."""
    W, H = args.leg_width, args.leg_height
    per_gpu = 625
    total = per_gpu * world
    raw, _, _ = ptk.scene_constants(os.path.join(ROOT, "data", "scenes", "cornell_crystal.scene"),
                                    W, H, args.depth, 0.005, False)
    ctx.set_constants(raw)
    ctx.set_stream_id(rank)
    planes = multigpu.device_planes(ctx, device)
    n_local = multigpu.shard_spp(total, world, rank)
    ctx.render(8, ptk.SEED_COUNTER)
    multigpu.all_reduce_(planes[0:3])
    ctx.reset()
    ctx.synchronize()
    _barrier(args, gpu)
    t0 = time.perf_counter()
    ctx.render(n_local, ptk.SEED_COUNTER)
    t1 = time.perf_counter()
    multigpu.all_reduce_(planes[0:3])
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    _barrier(args, gpu)
    elapsed = _max_over_ranks(time.perf_counter() - t0, device)
    return {"workload": "Scenes/cornell_crystal.scene %dx%d, %d spp split over %d GPUs, all-reduce of "
                        "the colour planes" % (W, H, total, world),
            "msamples_per_s": float(W) * H * total / elapsed / 1e6, "frame_ms": elapsed * 1e3,
            "render_ms_rank0": (t1 - t0) * 1e3, "exchange_ms_rank0": (t2 - t1) * 1e3,
            "exchange_bytes_per_rank": multigpu.exchange_bytes(W, H, world, features=False)}


def config5_leg(ctx, args, rank, world, gpu, device):
    """BASELINE.json configs[4]: Box_Grid 3840x2160 with features, samples split across the
    ranks (313 per GPU: 2500 spp on eight), band-wise merge (all-to-all, halo rows included),
    row-band denoiser N = 1 with BEST.txt, all-gather of the filtered bands."""
    W, H = args.leg_width, args.leg_height
    per_gpu = 313
    total = min(2500, per_gpu * world) if world == 8 else per_gpu * world
    raw, _, _ = ptk.scene_constants(os.path.join(ROOT, "data", "scenes", "Box_Grid.scene"),
                                    W, H, args.depth, 0.005, True)
    w = ptk.load_weights(os.path.join(ROOT, "data", "weights", "BEST.txt"))
    ctx.set_constants(raw)
    ctx.set_stream_id(rank)
    planes = multigpu.device_planes(ctx, device)
    n_local = multigpu.shard_spp(total, world, rank)
    out = {}
    for attempt in ("warm", "timed"):
        ctx.reset()
        ctx.synchronize()
        _barrier(args, gpu)
        t0 = time.perf_counter()
        ctx.render(n_local if attempt == "timed" else 4, ptk.SEED_COUNTER)
        t1 = time.perf_counter()
        n_total, lo, hi = multigpu.merge_to_bands(planes, n_local if attempt == "timed" else 4,
                                                  W, H, halo=3, ctx=ctx)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        den = multigpu.denoise_sharded(ctx, w, 1, device, n_total=n_total)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        band_ms = ctx.timing().denoise_ms
        _barrier(args, gpu)
        elapsed = _max_over_ranks(time.perf_counter() - t0, device)
        post = _max_over_ranks(t3 - t1, device)
        out = {"workload": "Scenes/Box_Grid.scene %dx%d with features, %d spp split over %d GPUs, "
                           "band-wise merge + row-band denoise N=1 BEST.txt + gather" % (W, H, total, world),
               "msamples_per_s": float(W) * H * total / elapsed / 1e6, "frame_ms": elapsed * 1e3,
               "render_ms_rank0": (t1 - t0) * 1e3,
               "denoise_frame_ms": post * 1e3,          # merge + band denoise + gather, max over ranks
               "merge_ms_rank0": (t2 - t1) * 1e3, "band_denoise_and_gather_ms_rank0": (t3 - t2) * 1e3,
               "band_denoise_kernels_ms_rank0": band_ms,
               "merge_bytes_received_per_rank": multigpu.exchange_bytes(W, H, world, halo=3),
               "gather_bytes_received_per_rank": int((world - 1) * 16 * W * (H // world + 1))}
    # After the clock has stopped: does every rank hold the same gathered frame?  Two checksums
    # of the denoised frame (a plain sum and one weighted by position, so that a band in the
    # wrong place shows), maximum and minimum over the ranks; each rank's own band is its own
    # output, so equal frames are right frames.
    flat = torch.nan_to_num(den.reshape(-1).double())
    pos = (torch.arange(flat.numel(), device=flat.device) % 65521).double()
    sums = torch.stack([flat.sum(), (flat * pos).sum()])
    both = torch.cat([sums, -sums])
    multigpu.all_reduce_(both, op=dist.ReduceOp.MAX)
    out["gathered_frames_agree"] = bool((both[0] == -both[2]) and (both[1] == -both[3]))
    out["gathered_frame_mean"] = float(both[0].item()) / flat.numel()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--spp-per-step", type=int, default=200)
    ap.add_argument("--scene", default="cornell_crystal")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--math", choices=["exact", "fast"], default="exact")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo rehearses the multi-rank path with several ranks on one GPU")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the multi-rank exchange path even with one rank (rehearsal)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-denoise-probe", action="store_true")
    ap.add_argument("--no-denoise-4k", action="store_true",
                    help="skip the 3840x2160 denoiser probe (profiling runs: keeps one frame size per kernel)")
    ap.add_argument("--no-multi-gpu-legs", action="store_true",
                    help="skip the 3840x2160 config 4 / config 5 legs of a multi-rank run")
    ap.add_argument("--leg-timeout", type=int, default=900,
                    help="seconds the multi-rank legs may take before the headline is printed without them")
    ap.add_argument("--leg-width", type=int, default=3840)
    ap.add_argument("--leg-height", type=int, default=2160)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libptk has no CPU path")
    # stdout carries the result line alone: whatever libraries print on the way (the collective
    # library writes a version banner to file descriptor 1) is sent to stderr instead
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    gpu = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(gpu)
    device = torch.device("cuda", gpu)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    W, H = args.width, args.height
    ctx = ptk.Context(gpu)
    raw, _, _ = ptk.scene_constants(os.path.join(ROOT, "data", "scenes", args.scene + ".scene"),
                                    W, H, args.depth, 0.005, False)
    ctx.set_constants(raw)
    ctx.set_math_mode(ptk.MATH_FAST if args.math == "fast" else ptk.MATH_EXACT)
    ctx.set_stream_id(rank)
    planes = multigpu.device_planes(ctx, device) if use_dist else None

    spp = args.spp_per_step
    kernel_ms = []
    kernel_rays = []

    def step():
        ctx.render(spp, ptk.SEED_COUNTER)     # returns when the launch has finished
        t = ctx.timing()
        kernel_ms.append(t.render_ms)
        kernel_rays.append(t.render_rays)

    def exchange():
        # the frame's one accumulation exchange: colour sums over ranks, in place in HBM
        if use_dist:
            multigpu.all_reduce_(planes[0:3])

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier(device_ids=[gpu]) if args.backend == "nccl" else dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    exchange()                                # warms the collective up as well
    ctx.reset()
    kernel_ms.clear()
    kernel_rays.clear()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    exchange()
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        multigpu.all_reduce_(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    samples_total = float(W) * H * spp * args.steps * world
    value = samples_total / elapsed / 1e6
    out = {
        "metric": "Msamples/sec (paths/s) cornell_crystal 1080p",
        "value": value, "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {
            "workload": "Scenes/%s.scene %dx%d, %d spp per step x %d steps per GPU, render only, "
                        "maxDepth %d, device counter seeds, math=%s" % (
                            args.scene, W, H, spp, args.steps, args.depth, args.math),
            "samples_per_step_per_gpu": W * H * spp,
            "exchange": ("one all-reduce of the 3 colour planes (%d bytes) at the end of the frame, "
                         "inside the timed region" % (3 * W * H * 4)) if use_dist else "none",
        },
    }
    legs = {}
    if use_dist and not args.no_multi_gpu_legs:
        # A rank that fails inside a leg leaves the others waiting in the leg's collectives.
        # Two guards keep the result line: a watchdog that prints the headline as it stands
        # and ends the process when the legs overrun, and after each leg an agreement (MIN
        # over ranks of an ok flag) on whether to go on.
        import threading

        def overrun():
            if rank == 0:
                out.update(legs)
                out["legs_error"] = "the multi-rank legs did not finish within %d s" % args.leg_timeout
                print(json.dumps(out), file=result_out, flush=True)
            os._exit(0 if rank == 0 else 3)

        watchdog = threading.Timer(args.leg_timeout, overrun)
        watchdog.daemon = True
        watchdog.start()
        for name, fn in (("config2_split", config2_split_leg), ("config4", config4_leg),
                         ("config5", config5_leg)):
            try:
                legs[name] = fn(ctx, args, rank, world, gpu, device)
                ok = 1.0
            except Exception as exc:   # a leg must not cost the run its result line
                legs[name] = {"error": "%s: %s" % (type(exc).__name__, exc)}
                ok = 0.0
            flag = torch.tensor([ok], dtype=torch.float64, device=device)
            multigpu.all_reduce_(flag, op=dist.ReduceOp.MIN)
            if float(flag.item()) < 0.5:
                legs.setdefault("legs_error", "a rank failed in %s; the remaining legs were skipped" % name)
                break
        watchdog.cancel()
    if rank == 0:
        out.update(legs)
        avg_ms = sum(kernel_ms) / max(len(kernel_ms), 1)
        samples_per_launch = float(W) * H * spp
        try:
            ops = load_opcount(args.scene)
            flops = ops["flops_per_sample"]
            achieved = samples_per_launch * flops / (avg_ms * 1e-3) / 1e12
            out["roofline"] = {
                "bound": "valu", "kernel": "render_kernel", "achieved": achieved,
                "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_TFLOPS,
                "traffic": None, "avg_launch_ms": avg_ms,
                "grays_per_s": sum(kernel_rays) / max(sum(kernel_ms), 1e-9) / 1e6,
                "flops_per_sample": flops, "int_ops_per_sample": ops.get("int_ops_per_sample"),
                "algorithmic_bytes": 2 * 3 * 4 * W * H,   # colour planes read + written once
                "note": "fp32 VALU bound (not HBM, not MFMA): algorithmic flops per sample from "
                        "the op-counting build of the CPU restatement; a correctly rounded "
                        "divide / sqrt / sin counts as one flop and no operation may be fused, so "
                        "the FMA peak is an upper bound no exact-arithmetic kernel can approach",
            }
            prof = load_profile(int(samples_per_launch), False) if args.scene == "cornell_crystal" \
                else None
            if prof and "pmc_per_launch" in prof and \
                    abs(prof.get("avg_ms", 0.0) / avg_ms - 1.0) > PROFILE_STALE_TOLERANCE:
                out["roofline"]["profile_note"] = (
                    "%s was taken on a launch of %.3f ms, this run's is %.3f ms: its counters are "
                    "not quoted (rerun scripts/collect_profiles.sh)" % (
                        os.path.basename(PROFILE_SUMMARY), prof.get("avg_ms", 0.0), avg_ms))
                prof = None
            if prof and "pmc_per_launch" in prof:
                pmc = prof["pmc_per_launch"]
                if "hbm_read_bytes_per_launch" in prof:
                    out["roofline"]["traffic"] = (prof["hbm_read_bytes_per_launch"]
                                                  + prof["hbm_write_bytes_per_launch"])
                    out["roofline"]["traffic_note"] = (
                        "rocprofv3 FETCH_SIZE (x calibration) + WRITE_SIZE per launch from "
                        "profiles/%s: register-spill scratch traffic, not plane traffic"
                        % os.path.basename(PROFILE_SUMMARY))
                if "SQ_INSTS_VALU" in pmc:
                    lane_instr = pmc["SQ_INSTS_VALU"] * 64.0
                    rate = lane_instr / (avg_ms * 1e-3) / 1e12
                    out["roofline"]["valu_issue"] = {
                        "lane_instructions_per_sample": lane_instr / samples_per_launch,
                        "achieved": rate, "peak": PEAK_VALU_ISSUE_T, "unit": "T lane-instr/s",
                        "frac": rate / PEAK_VALU_ISSUE_T,
                        "active_lane_frac": pmc.get("SQ_THREAD_CYCLES_VALU", 0.0)
                        / pmc["SQ_INSTS_VALU"] / 64.0,
                    }
        except (OSError, KeyError) as exc:
            out["roofline"] = {"bound": "valu", "achieved": None, "peak": PEAK_FP32_TFLOPS,
                               "unit": "TFLOP/s", "frac": None, "traffic": None,
                               "avg_launch_ms": avg_ms, "note": "op count unavailable: %s" % exc}
        if world == 1 and not args.force_dist:
            if not args.no_denoise_probe:
                out["denoise"] = denoise_probe(ctx, W, H, args.depth)
                if not args.no_denoise_4k:
                    out["denoise_4k"] = denoise_4k_probe(ctx, args.depth)
                out["scene_hierarchy"] = hierarchy_probe(ctx, W, H)
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(args.scene, W, H, args.depth)
        print(json.dumps(out), file=result_out, flush=True)
    ctx.close()
    if use_dist:
        dist.barrier(device_ids=[gpu]) if args.backend == "nccl" else dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
