"""Static instruction mix of the kernels in a hipcc -S listing: python scripts/asm_mix.py file.s [name-substring]."""
import collections
import re
import sys

def main():
    lines = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    name, ops = None, None
    for line in lines:
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, ops = m.group(1), collections.Counter()
            continue
        if name is None:
            continue
        t = line.strip()
        if t.startswith(".end_amdhsa_kernel") or t.startswith(".section"):
            if ops and want in name:
                groups = collections.Counter()
                for k, v in ops.items():
                    g = ("valu" if k.startswith("v_") else "salu" if k.startswith("s_") else "lds" if k.startswith("ds_")
                         else "vmem" if k.startswith(("global", "buffer", "flat", "scratch")) else "other")
                    groups[g] += v
                print(name, dict(groups))
                print("   ", ops.most_common(45))
            name = None
            continue
        m = re.match(r"^([a-z][a-z_0-9]+)(\s|$)", t)
        if m and not t.startswith((".", ";")):
            ops[m.group(1)] += 1

main()
