"""Writes the denoiser's outputs (denoised, wcSum) for a few frames to an .npz, with the library
named by PTK_LIB, so that two builds of the filter can be compared bit for bit
(scripts/gpu_filter_compare.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import numpy as np
import ptk
from gpu_denoise_probe import synthetic

def main():
    out = sys.argv[1]
    w = ptk.load_weights(os.path.join(ROOT, "data", "weights", "BEST.txt"))
    res = {}
    for (W, H, real) in ((1920, 1080, True), (1920, 1080, False), (253, 131, False), (64, 9, False), (61, 3, False), (3840, 2160, True)):
        ctx = ptk.Context(0)
        raw, _, _ = ptk.scene_constants(os.path.join(ROOT, "data", "scenes", "MultiLightGlass.scene"), W, H, 4, 0.005, True)
        ctx.set_constants(raw)
        if real:
            ctx.render(16, ptk.SEED_COUNTER)
        else:
            ctx.write_planes(synthetic(W, H), 128)
        r = ctx.denoise(w, 1, 0)
        tag = "%dx%d_%s" % (W, H, "real" if real else "noise")
        res[tag + "_den"] = r["denoised"]; res[tag + "_wc"] = r["wcSum"]
        # a band with ragged ends, on a cleared output
        if H > 40:
            ctx.denoise_rows(w, 1, 0, 7, H - 11)
            ctx.synchronize()
        ctx.close()
    np.savez(out, **res)

if __name__ == "__main__":
    main()
