"""Re-encodes images the reference's authors rendered and checked in under Renders/ (ASCII
PPM) as compressed arrays under tests/golden/reference_renders/: the noisy (-N) and denoised
(-D, -DNT) MultiLightGlass renders and the 384x216 EdgeBlur series (the cornell scene).  The
other files there (Comparisons/Ball-*, HalfWall-*) show scenes that are not among the shipped
.scene files and are left out.  Run where /root/reference exists."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/Renders"
DST = os.path.join(ROOT, "tests", "golden", "reference_renders")
os.makedirs(DST, exist_ok=True)
names = ["MultLightGlass-0.4-%d-N" % spp for spp in (5000, 2500, 128)]
names += ["MultLightGlass-0.4-%d-D" % spp for spp in (5000, 2500, 128)] + ["MultLightGlass-0.4-5000-DNT"]
names += ["EdgeBlur"] + ["EdgeBlur%d" % i for i in range(5)]
for name in names:
    t = open(os.path.join(SRC, name + ".ppm")).read().split()
    assert t[0] == "P3" and t[3] == "255"
    W, H = int(t[1]), int(t[2])
    rgb = np.array(t[4:], dtype=np.uint16).reshape(H, W, 3)
    np.savez_compressed(os.path.join(DST, name + ".npz"), rgb=rgb)
    print(name, rgb.shape, rgb.max())
