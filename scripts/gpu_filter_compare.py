"""Runs scripts/gpu_filter_dump.py with the default library and with every library under
path-tracer_amd/variants/, and reports whether the outputs are bit-identical to the first."""
import glob, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [""] + sorted(glob.glob(os.path.join(ROOT, "path-tracer_amd", "variants", "*.so")))
outs = []
for i, lib in enumerate(libs):
    env = dict(os.environ)
    if lib: env["PTK_LIB"] = lib
    out = os.path.join(ROOT, "gpurun_out", "fdump%d.npz" % i)
    subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gpu_filter_dump.py"), out], env=env, check=True, timeout=300)
    outs.append(out)
ref = np.load(outs[0])
for lib, out in zip(libs[1:], outs[1:]):
    d = np.load(out)
    for k in ref.files:
        a, b = ref[k], d[k]
        same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
        nd = int((a.view(np.uint32) != b.view(np.uint32)).sum())
        md = float(np.nanmax(np.abs(a - b) / (np.abs(a) + 1e-30))) if nd else 0.0
        print("%s vs default: %s %s (%d words differ, max rel %.3g)" % (os.path.basename(lib), k, "IDENTICAL" if same else "DIFFERENT", nd, md), flush=True)
for o in outs: os.remove(o)
