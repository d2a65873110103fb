"""Checks a gfx950 assembly listing (hipcc -S) for the hazards the compiler cannot see inside
inline-asm statements.  Before a DPP instruction,
  * a vector register it reads through its DPP operand must not have been written by a vector
    instruction fewer than two wait states earlier;
  * no operand of it may be the result of a transcendental instruction issued just before it
    (one wait state);
  * EXEC must not have been written by a vector instruction (v_cmpx_*, or any v_* whose
    destination is exec) fewer than five wait states earlier.
Every instruction is one wait state, `s_nop N` is N + 1.  The scan follows the control flow
backwards: at a label it continues both with the instruction above the label (unless that one
never falls through) and with the instructions above every branch to the label, the branch
itself counting one wait state, so a write just before a loop's back edge is paired with a DPP
read at the top of the loop.
Usage: check_dpp_hazard.py file.s   -- exits 1 and prints the offenders if any."""
import re
import sys

NEED_SRC, NEED_TRANS, NEED_EXEC = 2, 1, 5
TRANS = re.compile(r"v_(exp|log|rcp|rsq|sqrt|sin|cos)_")
NO_FALL_THROUGH = ("s_branch", "s_endpgm", "s_setpc_b64", "s_rfe_b64")


def regs(tok):
    tok = tok.split()[0] if tok.split() else tok
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def parse(path):
    """-> items: ("ins", line number, text) | ("label", line number, name)"""
    items = []
    for ln, line in enumerate(open(path), 1):
        t = line.split(";")[0].strip()
        if not t or t.startswith("."):
            m = re.fullmatch(r"(\.L\w+):", t)        # local labels start with a dot
            if m:
                items.append(("label", ln, m.group(1)))
            continue
        m = re.fullmatch(r"([\w.$]+):", t)
        if m:
            items.append(("label", ln, m.group(1)))
            continue
        items.append(("ins", ln, t))
    return items


def operands(text):
    op = text.split()[0]
    return op, [x.strip() for x in text[len(op):].split(",")]


def writes_exec(op, ops):
    if op.startswith("v_cmpx"):
        return True
    return op.startswith("v_") and bool(ops) and ops[0].split() and ops[0].split()[0] in ("exec", "exec_lo", "exec_hi")


def main(path):
    items = parse(path)
    branches = {}        # label -> indices of the branch instructions that target it
    for i, it in enumerate(items):
        if it[0] == "ins":
            op, ops = operands(it[2])
            if op.startswith("s_cbranch") or op == "s_branch":
                branches.setdefault(ops[0].split()[0], []).append(i)
    bad = set()
    checked = 0
    for i, it in enumerate(items):
        if it[0] != "ins":
            continue
        op, ops = operands(it[2])
        if not op.endswith("_dpp"):
            continue
        checked += 1
        ln = it[1]
        src = regs(ops[1])                           # the DPP operand is src0
        used = set()
        for o in ops[1:]:
            used |= regs(o)
        if op.startswith("v_fmac"):
            used |= regs(ops[0])
        seen = set()
        stack = [(i, 0)]                             # (index whose predecessors are examined, wait states so far)
        while stack:
            at, waited = stack.pop()
            if waited >= NEED_EXEC or (at, waited) in seen:
                continue
            seen.add((at, waited))
            j = at - 1
            if j < 0:
                continue
            kind, pl, pt = items[j]
            if kind == "label":
                for b in branches.get(pt, []):       # arriving through a branch: it costs one wait state
                    stack.append((b, waited + 1))
                k = j - 1                            # ... or by falling through from above
                while k >= 0 and items[k][0] == "label":
                    for b in branches.get(items[k][2], []):
                        stack.append((b, waited + 1))
                    k -= 1
                if k >= 0 and not items[k][2].split()[0].startswith(NO_FALL_THROUGH):
                    stack.append((k + 1, waited))
                continue
            pop, pops = operands(pt)
            if pop == "s_nop":
                stack.append((j, waited + int(pt.split()[1], 0) + 1))
                continue
            if pop.startswith("v_"):
                dst = regs(pops[0]) if pops else set()
                if waited < NEED_SRC and dst & src:
                    bad.add("%s:%d: %s reads v%s written at line %d (%s), %d wait state(s) before" % (
                        path, ln, op, sorted(dst & src), pl, pt, waited))
                if waited < NEED_TRANS and TRANS.match(pop) and dst & used:
                    bad.add("%s:%d: %s reads v%s from the transcendental at line %d" % (
                        path, ln, op, sorted(dst & used), pl))
                if writes_exec(pop, pops):
                    bad.add("%s:%d: %s follows a vector write of EXEC at line %d (%s) by %d wait state(s), 5 needed" % (
                        path, ln, op, pl, pt, waited))
            stack.append((j, waited + 1))
    for b in sorted(bad):
        print(b)
    print("%d DPP instructions checked, %d hazards" % (checked, len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
