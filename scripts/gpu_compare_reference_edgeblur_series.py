"""Renders/EdgeBlur0..4.ppm (384x216, the cornell scene) look like Renders/EdgeBlur.ppm after
ever stronger smoothing.  If they are the reference's denoiser at N = 0, 1, 2, ... with one of
the shipped weight files, each would be matched by that file at its N.  Renders cornell at 36
spp, depth 12 (what EdgeBlur.ppm fits), twice; denoises with every weight file and N = 0..8;
prints, per reference image, `pix` (mean |ours - file| per pixel, 8-bit levels), `self` (the
same between our two seeds) and `hf` (RMS of image - 5x5 box mean) for every combination.
Usage: gpu_compare_reference_edgeblur_series.py > profiles/r03_reference_edgeblur_series.txt"""
import glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import ptk
import reference_renders as RR

D = os.path.join(ROOT, "tests", "golden", "reference_renders")


def main():
    E = RR.EDGEBLUR
    W, H, spp = E["width"], E["height"], E["spp"]
    raw, _, _ = ptk.scene_constants(os.path.join(ROOT, "data", "scenes", "cornell.scene"), W, H, E["depth"], 0.005, True)
    ctxs = []
    for stream in (0, 1):
        c = ptk.Context(0)
        c.set_constants(raw)
        c.set_stream_id(stream)
        c.render(spp, ptk.SEED_COUNTER)
        ctxs.append(c)
    noisy = RR.edgeblur_file_order(ctxs[0].read_planes(0, 3), spp)
    e = RR.fit_exposure_blocks(noisy, RR.load_edgeblur())
    refs = {i: np.load(os.path.join(D, "EdgeBlur%d.npz" % i))["rgb"].astype(np.int32) for i in range(5)}
    print("exposure %.4f (fitted on EdgeBlur.ppm); hf of EdgeBlur.ppm %.3f, of EdgeBlur0..4: %s; channel means of 0..4: %s" % (
        e, RR.fine_structure(RR.load_edgeblur()), " ".join("%.3f" % RR.fine_structure(r) for r in refs.values()),
        " ".join(str(np.round(r.mean(axis=(0, 1)), 1)) for r in refs.values())))
    print("%-20s %2s %7s %7s | %s" % ("weights", "N", "self", "hf", " ".join("pix vs %d" % i for i in range(5))))
    for wf in sorted(glob.glob(os.path.join(ROOT, "data", "weights", "*.txt"))):
        w = ptk.load_weights(wf)
        for n in range(0, 9):
            q = []
            for c in ctxs:
                r = c.denoise(w, n, 0)
                q.append(RR.quantise(np.asarray(r["denoised"], np.float64).reshape(H, W, 3)[::-1], e))
            print("%-20s %2d %7.3f %7.3f | %s" % (os.path.basename(wf), n, np.abs(q[0] - q[1]).mean(), RR.fine_structure(q[0]),
                                                 " ".join("%8.3f" % np.abs(q[0] - refs[i]).mean() for i in range(5))), flush=True)


if __name__ == "__main__":
    main()
