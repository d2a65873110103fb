#!/bin/bash
# Instruction-cache and scalar-cache counters of the kernels of a short bench run.
# Usage: scripts/pmc_icache.sh <tag>   (outputs under gpurun_out/<tag>/)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/${1:-icache}
mkdir -p "$OUT"
SHORT="--steps 2 --warmup 1 --spp-per-step 200 --no-cpu-baseline --no-denoise-probe"
timeout -s KILL 150 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_VALU SQ_BUSY_CYCLES \
    --output-format csv -d "$OUT/a" -- python3 bench.py $SHORT > "$OUT/a.json" 2> "$OUT/a.err" || { tail -5 "$OUT/a.err"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:72]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    if "render_kernel" not in k: continue
    print(k)
    for c, v in sorted(cs.items()):
        print("    %-22s %16.0f  (x%d)" % (c, sum(v) / len(v), len(v)))
PY
