#!/bin/bash
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/bvh3
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_max GRBM_GUI_ACTIVE --output-format csv -d "$OUT/ta" -- python3 scripts/gpu_bvh_perf.py 46 --no-extra > "$OUT/ta.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum --output-format csv -d "$OUT/tcp2" -- python3 scripts/gpu_bvh_perf.py 46 --no-extra > "$OUT/tcp2.log" 2>&1
timeout -s KILL 120 rocprofv3 --pmc TD_TD_BUSY_sum TD_BUSY_max GRBM_GUI_ACTIVE --output-format csv -d "$OUT/td" -- python3 scripts/gpu_bvh_perf.py 46 --no-extra > "$OUT/td.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(float); cnt = collections.defaultdict(int)
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "render_kernel" not in r["Kernel_Name"]: continue
        tot[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
for k in sorted(tot): print("%-34s total %.4g over %d records" % (k, tot[k], cnt[k]))
PY
