"""Counts the algorithmic operations per sample of the render path with the
op-counting build of the CPU restatement and writes profiles/opcount.json.

Run from the repository root:  python scripts/count_ops.py
"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

SEED = (0x9E3779B97F4A7C15, 0xD1B54A32D192ED03)


def main():
    here = os.path.join(ROOT, "oracle")
    subprocess.check_call(["make", "-C", here, "libptk_oracle_count.so"], stdout=subprocess.DEVNULL)
    L = C.CDLL(os.path.join(here, "libptk_oracle_count.so"))
    fp = C.POINTER(C.c_float)
    L.orc_render.argtypes = [C.POINTER(O.Constants), fp, C.c_int, C.POINTER(C.c_int),
                             C.POINTER(C.c_uint64), C.c_int, C.POINTER(C.c_uint64)]
    L.orc_render.restype = None
    table = {}
    W, H, spp = 1920, 1080, 4   # the benchmark's frame size
    for scene in ("cornell", "cornell_crystal", "MultiLightGlass", "Box_Grid"):
        sc = O.load_scene(os.path.join(ROOT, "data", "scenes", scene + ".scene"))
        for feats in (False, True):
            consts = O.build_constants(sc, W, H, 4, 0.005, feats)
            acc = O.new_accumulators(W, H)
            L.orc_opcount_reset()
            n = C.c_int(0)
            st = (C.c_uint64 * 2)(*SEED)
            rays = C.c_uint64(0)
            L.orc_render(C.byref(consts), acc.ctypes.data_as(fp), spp, C.byref(n), st, 1,
                         C.byref(rays))
            out = (C.c_uint64 * 5)()
            L.orc_opcount_get(out)
            flops, iops, libm, tests, samples = [int(x) for x in out]
            key = scene + ("+features" if feats else "")
            table[key] = {
                "flops_per_sample": flops / samples, "int_ops_per_sample": iops / samples,
                "libm_calls_per_sample": libm / samples, "shape_tests_per_sample": tests / samples,
                "config": "%dx%d, %d spp, maxDepth 4, randSamp 0.005" % (W, H, spp),
            }
            print(key, table[key])
    with open(os.path.join(ROOT, "profiles", "opcount.json"), "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
