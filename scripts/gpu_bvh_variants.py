"""scripts/gpu_bvh_perf.py for each libptk variant in path-tracer_amd/variants (and the default)."""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for lib in sorted(glob.glob(os.path.join(ROOT, "path-tracer_amd", "variants", "*.so"))) + [""]:
    print("==", os.path.basename(lib) or "libptk.so (default)", flush=True)
    env = dict(os.environ)
    if lib:
        env["PTK_LIB"] = lib
    subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gpu_bvh_perf.py")] + (sys.argv[1:] or ["22", "46"]) + ["--no-extra"],
                   env=env, timeout=300)
