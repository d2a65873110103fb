"""Renders every shipped scene at 384x216 (64 spp, depth 4 and 12) and writes tone-mapped PNGs
under gpurun_out/gallery/: a look at which scene an image of the reference's Renders/ shows."""
import glob, os, struct, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "path-tracer_amd"))
import numpy as np
import ptk


def write_png(path, img):
    img = np.clip(img, 0, 255).astype(np.uint8)
    H, W, _ = img.shape
    raw = b"".join(b"\x00" + img[y].tobytes() for y in range(H))
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", W, H, 8, 2, 0, 0, 0))
                           + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def main():
    out = os.path.join(ROOT, "gpurun_out", "gallery")
    os.makedirs(out, exist_ok=True)
    ctx = ptk.Context(0)
    W, H = 384, 216
    for path in sorted(glob.glob(os.path.join(ROOT, "data", "scenes", "*.scene"))):
        name = os.path.basename(path)[:-6]
        for depth in (4, 12):
            raw, _, _ = ptk.scene_constants(path, W, H, depth, 0.005, False)
            ctx.set_constants(raw)
            ctx.render(64, ptk.SEED_COUNTER)
            img = ctx.read_planes(0, 3).reshape(3, H, W)[:, ::-1].transpose(1, 2, 0) / 64.0
            e = 9.6 * float((img * [0.2125, 0.7154, 0.0721]).sum(axis=2).mean()) / 4.0
            write_png(os.path.join(out, "%s_d%d.png" % (name, depth)), 255.999 * (np.maximum(img, 0) / max(e, 1e-6)) ** (1 / 2.1))
            print(name, depth, img.mean(axis=(0, 1)).round(4), flush=True)


if __name__ == "__main__":
    main()
