"""Can the reference's checked-in DENOISED renders pin the denoiser?

Renders/MultLightGlass-0.4-{128,2500,5000}-D.ppm and -5000-DNT.ppm are outputs of the
reference's denoiser (Denoiser.cpp / DenoiserNN.cpp, or their CUDA copies: the GUI's default
denoising backend is 2, PT.cpp:58) on renders like the -N files.  Their names record neither
the weight file nor the window half-width N (a free integer in the GUI, PT.cpp:425).  This
script renders MultiLightGlass 768x432 twice (two seed streams) at each sample count with the
depth the -N file fitted, denoises both with every shipped weight file and N = 1..8, applies
the post-process with the exposure fitted on the -N file, and compares with the -D file:

  pix   mean |ours - file| per pixel and channel, in 8-bit levels
  self  the same between our two independently seeded results (the noise floor of `pix`)
  blk   mean |difference| of 16x16 block means (bias)
  hf    RMS of (image - its 5x5 box mean): how much fine structure / noise is left; ours | file

A combination pins the denoiser if pix is near self AND hf matches for all three sample counts.
Writes the table to stdout; run with > profiles/r03_reference_denoised_scan.txt.
Usage: gpu_compare_reference_denoised.py [max N]"""
import glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from scipy.ndimage import uniform_filter
import ptk
import reference_renders as RR

D = os.path.join(ROOT, "tests", "golden", "reference_renders")
W, H = RR.WIDTH, RR.HEIGHT


def file_order_mean(pixel_major):
    """(P, 3) per-sample means, bottom row first -> (H, W, 3) in the file's row order."""
    return np.asarray(pixel_major, np.float64).reshape(H, W, 3)[::-1]


def hf(img):
    img = img.astype(np.float64)
    return float(np.sqrt(np.mean((img - uniform_filter(img, size=(5, 5, 1), mode="nearest")) ** 2)))


def main():
    max_n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    ctx = ptk.Context(0)
    scene = os.path.join(ROOT, "data", "scenes", "MultiLightGlass.scene")
    weight_files = sorted(glob.glob(os.path.join(ROOT, "data", "weights", "*.txt")))
    best = {}
    for spp, depth in sorted(RR.FILES.items()):
        noisy_ref = RR.load(spp)
        targets = [("D", np.load(os.path.join(D, "MultLightGlass-0.4-%d-D.npz" % spp))["rgb"].astype(np.int32))]
        if spp == 5000:
            targets.append(("DNT", np.load(os.path.join(D, "MultLightGlass-0.4-5000-DNT.npz"))["rgb"].astype(np.int32)))
        raw, _, _ = ptk.scene_constants(scene, W, H, depth, 0.005, True)
        ctxs = []
        for stream in (0, 1):
            c = ptk.Context(0)
            c.set_constants(raw)
            c.set_stream_id(stream)
            c.render(spp, ptk.SEED_COUNTER)
            ctxs.append(c)
        e = RR.fit_exposure(RR.file_order(ctxs[0].read_planes(0, 3), spp), noisy_ref)
        qn = RR.quantise(RR.file_order(ctxs[0].read_planes(0, 3), spp), e)
        print("== %d spp, depth %d, exposure %.4f (fitted on the -N file); hf of the -N file %.3f, of our "
              "noisy render %.3f; -N file vs -D file: pix %.3f" % (
                  spp, depth, e, hf(noisy_ref), hf(qn), RR.pixel_stats(noisy_ref, targets[0][1])[0]), flush=True)
        for tag, ref in targets:
            print("-- against MultLightGlass-0.4-%d-%s (hf %.3f)" % (spp, tag, hf(ref)))
            print("%-20s %2s %8s %8s %8s %8s" % ("weights", "N", "pix", "self", "blk", "hf ours"))
            for wf in weight_files:
                w = ptk.load_weights(wf)
                for n in range(1, max_n + 1):
                    q = []
                    for c in ctxs:
                        r = c.denoise(w, n, 0)
                        q.append(RR.quantise(file_order_mean(r["denoised"]), e))
                    pix = RR.pixel_stats(q[0], ref)[0]
                    slf = RR.pixel_stats(q[0], q[1])[0]
                    blk = RR.block_diff(q[0], ref)
                    row = (os.path.basename(wf), n, pix, slf, blk, hf(q[0]))
                    print("%-20s %2d %8.3f %8.3f %8.3f %8.3f" % row, flush=True)
                    key = (spp, tag)
                    if key not in best or pix < best[key][2]:
                        best[key] = row
        for c in ctxs:
            c.close()
    print("== closest combination per file (by pix)")
    for (spp, tag), row in sorted(best.items()):
        print("%d-%s: %s N=%d pix %.3f self %.3f blk %.3f hf %.3f" % ((spp, tag) + row))


if __name__ == "__main__":
    main()
