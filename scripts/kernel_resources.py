"""Registers, scratch and LDS of every kernel in hipcc -S listings:
python scripts/kernel_resources.py file.s [file.s ...] [> profiles/rNN_kernel_resources.txt]"""
import re
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
        return [re.sub(r"\(anonymous namespace\)::|^void ", "", o) for o in out]
    except OSError:
        return names


def main():
    for path in sys.argv[1:]:
        txt = open(path).read()
        rows = []
        for block in txt.split("  - .agpr_count:")[1:]:
            block = block.split("\n  - .agpr_count:")[0]
            name = re.search(r"\.name:\s+(\S+)", block).group(1)
            g = lambda k: int(re.search(k + r":\s+(\d+)", block).group(1))
            rows.append((name, g(r"\.vgpr_count"), g(r"\.sgpr_count"), g(r"\.private_segment_fixed_size"),
                         g(r"\.vgpr_spill_count"), g(r"\.group_segment_fixed_size")))
        names = demangle([r[0] for r in rows])
        print("== %s" % path.split("/")[-1])
        print("%-64s %5s %5s %8s %7s %7s" % ("kernel", "vgpr", "sgpr", "scratch", "spilled", "lds"))
        for r, n in zip(rows, names):
            print("%-64s %5d %5d %8d %7d %7d" % ((re.sub(r"\(RenderArgs\)|\(DenoiseArgs.*\)", "", n)[:64],) + r[1:]))


main()
