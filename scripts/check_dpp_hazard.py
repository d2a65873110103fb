"""Checks a gfx950 assembly listing (hipcc -S) for the hazard the compiler cannot see inside
inline-asm statements: a DPP instruction must not read, through its DPP operand, a vector
register written by one of the two instructions before it (two wait states; `s_nop N` counts
N + 1); nor may it read any operand written by the transcendental instruction just before it
(one wait state).  Usage: check_dpp_hazard.py file.s   -- exits 1 and prints the offenders if any."""
import re, sys

def regs(tok):
    m = re.fullmatch(r"v(\d+)", tok)
    if m: return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()

def main(path):
    ins = []
    for ln, line in enumerate(open(path), 1):
        t = line.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"): continue
        ins.append((ln, t))
    bad = 0
    for i, (ln, t) in enumerate(ins):
        op = t.split()[0]
        if not op.endswith("_dpp"): continue
        ops = [x.strip() for x in t[len(op):].split(",")]
        src = regs(ops[1].split()[0])            # the DPP operand is src0
        if op.startswith("v_mov_b32"): src = regs(ops[1].split()[0])
        if i > 0:
            pl, pt = ins[i - 1]
            pop = pt.split()[0]
            if re.match(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_", pop):
                dst = regs(pt[len(pop):].split(",")[0].strip())
                used = set()
                for o in ops[1:]: used |= regs(o.split()[0])
                if op.startswith("v_fmac"): used |= regs(ops[0].split()[0])
                if dst & used:
                    print("%s:%d: %s reads v%s from the transcendental at line %d" % (path, ln, op, sorted(dst & used), pl))
                    bad += 1
        wait = 0
        for j in range(i - 1, max(i - 4, -1), -1):
            pl, pt = ins[j]
            pop = pt.split()[0]
            if wait >= 2: break
            if pop == "s_nop":
                wait += int(pt.split()[1], 0) + 1
                continue
            if pop.startswith("v_"):
                dst = regs(pt[len(pop):].split(",")[0].strip())
                if dst & src:
                    print("%s:%d: %s reads v%s written at line %d (%s)" % (path, ln, op, sorted(dst & src), pl, pt))
                    bad += 1
            wait += 1
    print("%d DPP instructions checked, %d hazards" % (sum(1 for _, t in ins if t.split()[0].endswith("_dpp")), bad))
    return 1 if bad else 0

if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
