"""Host code under AddressSanitizer and UndefinedBehaviorSanitizer: the facade's scene and weight
file parsers, scene validation, shape compilation, the hierarchy builder and its verifier, the
band partition and the argument checks of the C ABI, driven by tests/host_sanitize_main.cpp over
damaged copies of every shipped scene and weight file and over random shape tables (with NaN,
infinities, out-of-range indices and sizes).  Only host code is instrumented (device code cannot
be on this image); the kernels' own objects are linked in as built.  No GPU is needed."""
import os
import random
import shutil
import subprocess

import pytest

from conftest import ROOT

HIPCC = "/opt/rocm/bin/hipcc"
PKG = os.path.join(ROOT, "path-tracer_amd")
SAN = {"memory": (["-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined",
                  "-Xarch_host", "-fno-sanitize-recover=undefined", "-Xarch_host", "-fno-omit-frame-pointer"],
                 ["-fsanitize=address", "-fsanitize=undefined", "-fno-sanitize-recover=undefined"]),
       "thread": (["-Xarch_host", "-fsanitize=thread", "-Xarch_host", "-fno-omit-frame-pointer"],
                  ["-fsanitize=thread"])}
HOST_SOURCES = ["ptk_api.cpp", "bvh_build.cpp"]        # every line of host logic of libptk.so


def _build(out_dir, flavour="memory"):
    """libptk with its host translation units instrumented + the driver; returns the binary."""
    compile_san, link_san = SAN[flavour]
    objs = []
    if not os.path.isdir(os.path.join(PKG, "build")) or len(os.listdir(os.path.join(PKG, "build"))) < 10:
        subprocess.run(["make", "-C", PKG, "-j8", "libptk.so"], check=True, timeout=3000)   # a fresh tree
    for name in sorted(os.listdir(os.path.join(PKG, "build"))):
        if name.endswith(".o") and name[:-2] not in HOST_SOURCES:
            objs.append(os.path.join(PKG, "build", name))
    assert len(objs) >= 10, "build the library first (make -C path-tracer_amd)"
    flags = ["-O1", "-g", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off",
             "-fno-fast-math", "-w", "-I" + os.path.join(ROOT, "include")]
    for src in HOST_SOURCES:
        obj = os.path.join(out_dir, src + ".o")
        subprocess.run([HIPCC] + flags + compile_san + ["-x", "hip", "-c", os.path.join(PKG, "csrc", src),
                                                 "-o", obj], check=True, timeout=900)
        objs.append(obj)
    exe = os.path.join(out_dir, "host_sanitize")
    subprocess.run([HIPCC, "-O1", "-g", "-std=c++17", "--offload-arch=gfx950", "-w"] + link_san +
                   ["-fno-gpu-sanitize", "-I" + os.path.join(ROOT, "include"),
                    "-x", "c++", os.path.join(ROOT, "tests", "host_sanitize_main.cpp"),
                    os.path.join(PKG, "host", "ptk_host.cpp"), "-x", "none"] + objs + ["-o", exe],
                   check=True, timeout=900)
    return exe


def _damaged(text, rng):
    """One damaged copy of a text file: the kinds of damage a hand-edited or truncated file has."""
    lines = text.split("\n")
    kind = rng.randrange(9)
    if kind == 0:
        return text[:rng.randrange(len(text) + 1)]
    if kind == 1 and lines:
        del lines[rng.randrange(len(lines))]
    elif kind == 2 and lines:
        i = rng.randrange(len(lines))
        lines[i] = lines[i].replace("/", " ", rng.randrange(1, 4))
    elif kind == 3 and lines:
        i = rng.randrange(len(lines))
        tok = lines[i].split(" ")
        tok[rng.randrange(len(tok))] = rng.choice(["nan", "inf", "-inf", "1e999", "-1", "", "0x7fffffff",
                                                    "99999999999999999999", "abc", "4294967295", "1e-50"])
        lines[i] = " ".join(tok)
    elif kind == 4:
        lines = lines * rng.randrange(2, 9)                 # far beyond the fixed-size limits
    elif kind == 5 and lines:
        i = rng.randrange(len(lines))
        lines[i] = lines[i][:1] + lines[i][1:] * rng.randrange(2, 40)
    elif kind == 6:
        chars = list(text)
        for _ in range(rng.randrange(1, 30)):
            if chars:
                chars[rng.randrange(len(chars))] = chr(rng.randrange(1, 256))
        return "".join(chars)
    elif kind == 7:
        lines.insert(rng.randrange(len(lines) + 1), rng.choice(["o 7", "o -1", "i 999", "i -4", "d", "m", "c",
                                                                 "pp", "o", "i 2147483648", "d 1/2", "m 1/1"]))
    else:
        rng.shuffle(lines)
    return "\n".join(lines)


@pytest.fixture(scope="module")
def sanitize_exe(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    return _build(str(tmp_path_factory.mktemp("san")))


def _run(exe, args, tmp_path):
    env = dict(os.environ, PTK_SANITIZE_SCRATCH=str(tmp_path),
               ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:verify_asan_link_order=0",
               UBSAN_OPTIONS="print_stacktrace=1", TSAN_OPTIONS="halt_on_error=1:exitcode=66",
               OMP_NUM_THREADS="4")
    return subprocess.run([exe] + args, capture_output=True, text=True, timeout=900, env=env,
                          errors="replace")


def test_parsers_survive_damaged_files(sanitize_exe, tmp_path):
    rng = random.Random(20240611)
    files = []
    for sub, ext in (("scenes", ".scene"), ("weights", ".txt")):
        src_dir = os.path.join(ROOT, "data", sub)
        for name in sorted(os.listdir(src_dir)):
            if not name.endswith(ext):
                continue
            text = open(os.path.join(src_dir, name), errors="replace").read()
            intact = tmp_path / ("intact_" + name)
            shutil.copy(os.path.join(src_dir, name), intact)
            files.append(str(intact))
            for k in range(24):
                p = tmp_path / ("%s_%02d%s" % (name[:-len(ext)], k, ext))
                p.write_text(_damaged(text, rng), errors="replace")
                files.append(str(p))
    for extra, body in (("empty.scene", ""), ("empty.txt", ""), ("nul.scene", "o 0\nd \0\0\0 1 0\n"),
                        ("long.scene", "d " + "1/" * 200000 + "\n"), ("missing.scene", None)):
        p = tmp_path / extra
        if body is not None:
            p.write_text(body)
        files.append(str(p))
    r = _run(sanitize_exe, files, tmp_path)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    parsed, accepted = [int(t) for t in r.stdout.split() if t.isdigit()][-2:]
    assert parsed == len(files) and accepted >= 18          # every intact file at least


def test_scene_tables_with_poison_are_refused_or_built(sanitize_exe, tmp_path):
    r = _run(sanitize_exe, ["--tables", "240"], tmp_path)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]
    assert "tables: 240 sets" in r.stdout


def test_hierarchy_builder_threads_under_thread_sanitizer(tmp_path):
    """The builder shares ranges of shapes between host threads above 32768 shapes (every 40th
    table set of the driver has 40 000): the same run with ThreadSanitizer on the host code."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    exe = _build(str(tmp_path), "thread")
    r = _run(exe, ["--tables", "80"], tmp_path)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert "tables: 80 sets" in r.stdout
