#!/bin/bash
# Runs on the GPU box: PMC passes over the ptk_set_scene (hierarchy) path-trace kernel.
# Usage: scripts/pmc_bvh.sh <tag> [grid side]
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-bvh}
SIDE=${2:-46}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES \
    --output-format csv -d "$OUT/sq" -- python3 scripts/gpu_bvh_perf.py $SIDE --no-extra > "$OUT/sq.log"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES SQ_INSTS_FLAT SQ_INSTS_SMEM SQ_INSTS_LDS \
    --output-format csv -d "$OUT/sq2" -- python3 scripts/gpu_bvh_perf.py $SIDE --no-extra > "$OUT/sq2.log"
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum \
    --output-format csv -d "$OUT/tcp" -- python3 scripts/gpu_bvh_perf.py $SIDE --no-extra > "$OUT/tcp.log"
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$OUT/tcc" -- \
    python3 scripts/gpu_bvh_perf.py $SIDE --no-extra > "$OUT/tcc.log"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- \
    python3 scripts/gpu_bvh_perf.py $SIDE --no-extra > "$OUT/fetch.log"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "render_kernel" not in k:
            continue
        tot[r["Counter_Name"]]["v"] += float(r["Counter_Value"])
        cnt[r["Counter_Name"]] += 1
for k in sorted(tot):
    print("%-34s total %.4g over %d records" % (k, tot[k]["v"], cnt[k]))
PY
