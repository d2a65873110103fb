import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for lib in sorted(glob.glob(os.path.join(ROOT, "path-tracer_amd", "variants", "*.so"))) + [""]:
    print("==", os.path.basename(lib) or "libptk.so (default)", flush=True)
    env = dict(os.environ)
    if lib: env["PTK_LIB"] = lib
    subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gpu_denoise_probe.py")] + (sys.argv[2:4] if len(sys.argv) > 3 else ["1920", "1080"]) + ["10"] + sys.argv[1:2], env=env, timeout=200)
