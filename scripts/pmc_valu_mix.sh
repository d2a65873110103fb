#!/bin/bash
# Dynamic VALU instruction mix of every kernel of a short bench run (two PMC passes, SQ block).
# Usage: scripts/pmc_valu_mix.sh <tag>   (outputs under gpurun_out/<tag>/)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/${1:-valu_mix}
mkdir -p "$OUT"
SHORT="--steps 2 --warmup 1 --spp-per-step 200 --no-cpu-baseline"
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT \
    --output-format csv -d "$OUT/a" -- python3 bench.py $SHORT > "$OUT/a.json" 2> "$OUT/a.err" || exit 1
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES \
    --output-format csv -d "$OUT/b" -- python3 bench.py $SHORT > "$OUT/b.json" 2> "$OUT/b.err" || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in "ab":
    for f in glob.glob(sys.argv[1] + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:72]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if "rocclr" in k: continue
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    tot = m.get("SQ_INSTS_VALU", 0.0)
    if tot <= 0: continue
    print(k)
    named = 0.0
    for c in sorted(m):
        if c.startswith("SQ_INSTS_VALU_"):
            named += m[c]
            print("    %-26s %14.0f  %5.1f %%" % (c[14:], m[c], 100 * m[c] / tot))
    print("    %-26s %14.0f  %5.1f %%   (selects, compares, moves, min/max, DPP ...)" % ("other", tot - named, 100 * (tot - named) / tot))
    for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CU_CYCLES"):
        if c in m: print("    %-26s %14.0f" % (c, m[c]))
PY
