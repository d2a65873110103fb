"""Regenerates tests/golden/*.npz from the CPU oracle (canonical variant F, plus the
render results of variant R: cos/sin/acos/pow correctly rounded, which an implementation
that rounds those four calls correctly must reproduce bit for bit).

The reference ships no golden vectors and cannot be built in this image, so
these fixtures are outputs of the pinned oracle (see tests/test_oracle_pins.py
for what pins it).  They freeze the integer RNG streams bit-exactly and the
floating-point stages as produced on x86-64 glibc, so that a drift in the
oracle, its compiler flags or the host libm shows up as a test failure.

Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
SEED = (0x9E3779B97F4A7C15, 0xD1B54A32D192ED03)
SCENES = ["cornell", "cornell_crystal", "MultiLightGlass", "Box_Grid", "blur_test", "bars3"]


def one_sample_per_pixel(consts, pix, W, H):
    """(P, 17): xyz, normal, albedo1, albedo2, worldPos, directLight, rays of one sample."""
    samples = np.zeros((H * W, 17), dtype=np.float32)
    for idx in range(W * H):
        s = O.render_sample(consts, idx // W, idx % W, int(pix[idx, 0]), int(pix[idx, 1]))
        samples[idx, 0:3] = s.xyz
        samples[idx, 3:6] = s.normal
        samples[idx, 6:9] = s.albedo1
        samples[idx, 9:12] = s.albedo2
        samples[idx, 12:15] = s.world_pos
        samples[idx, 15] = s.direct_light
        samples[idx, 16] = s.rays_sent
    return samples


def main():
    # (i) RNG known answers: first 64 pixel seeds, first 64 draws of pixel 0
    seeds, state = O.seed_stream(SEED, 64)
    draws, _ = O.rand_stream(int(seeds[0, 0]), int(seeds[0, 1]), 64)
    draws_pm1, _ = O.rand_stream(int(seeds[1, 0]), int(seeds[1, 1]), 64, -1.0, 1.0)
    np.savez(os.path.join(OUT, "rng_kat.npz"), seed=np.array(SEED, dtype=np.uint64),
             pixel_seeds=seeds, state_after=np.array(state, dtype=np.uint64),
             draws01_bits=draws.view(np.uint32), draws_pm1_bits=draws_pm1.view(np.uint32))

    # (ii) one-sample results per pixel, (iii) accumulators after 8 spp, (iv) exposure
    W, H = 64, 36
    for name in SCENES:
        sc = O.load_scene(os.path.join(ROOT, "data", "scenes", name + ".scene"))
        consts = O.build_constants(sc, W, H, 4, 0.005, True)
        pix, _ = O.seed_stream(SEED, W * H)
        samples = one_sample_per_pixel(consts, pix, W, H)
        acc = O.new_accumulators(W, H)
        n, glo, rays = O.render(consts, acc, 8, 0, SEED, threads=4)
        exps = [O.auto_exposure(acc, W, H, n, f)[0] for f in (0, 1, 2)]
        raw = O.auto_exposure(acc, W, H, n, 0)[1]
        with O.variant("R"):
            samples_r = one_sample_per_pixel(consts, pix, W, H)
            acc_r = O.new_accumulators(W, H)
            n_r, glo_r, rays_r = O.render(consts, acc_r, 8, 0, SEED, threads=4)
            assert n_r == n and glo_r == glo
        np.savez_compressed(os.path.join(OUT, "render_%s_64x36.npz" % name),
                            constants=np.frombuffer(bytes(consts), dtype=np.uint8),
                            samples_1spp=samples, acc_8spp=acc, sample_count=n,
                            samples_1spp_R=samples_r, acc_8spp_R=acc_r, rays_last_R=rays_r,
                            glo_after=np.array(glo, dtype=np.uint64), rays_last=rays,
                            exposure=np.array(exps, dtype=np.float32), raw_sum=raw)

    # (v) denoiser stages, MultiLightGlass 128x72, 8 spp, BEST.txt, N in {1,3}
    W, H = 128, 72
    sc = O.load_scene(os.path.join(ROOT, "data", "scenes", "MultiLightGlass.scene"))
    consts = O.build_constants(sc, W, H, 4, 0.005, True)
    acc = O.new_accumulators(W, H)
    n, glo, rays = O.render(consts, acc, 8, 0, SEED, threads=4)
    w = O.load_weights(os.path.join(ROOT, "data", "weights", "BEST.txt"))
    feats, var = O.forward_prop(acc, W, H, n, w)
    d1, wc1 = O.filter(acc, var, W, H, n, 1)
    d3, wc3 = O.filter(acc, var, W, H, n, 3)
    np.savez_compressed(os.path.join(OUT, "denoise_MultiLightGlass_128x72.npz"),
                        acc_8spp=acc, sample_count=n, features36=feats, variances7=var,
                        denoised_n1=d1, wcsum_n1=wc1, denoised_n3=d3, wcsum_n3=wc3)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
