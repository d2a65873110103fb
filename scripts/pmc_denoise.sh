#!/bin/bash
# PMC passes over the denoiser kernels alone (synthetic accumulators, 1080p).
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/${1:-pmc_dn}
KIND=${2:-synthetic}
mkdir -p "$OUT"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
    --output-format csv -d "$OUT/a" -- python3 scripts/gpu_denoise_probe.py 1920 1080 4 $KIND > "$OUT/a.log"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES \
    --output-format csv -d "$OUT/b" -- python3 scripts/gpu_denoise_probe.py 1920 1080 4 $KIND > "$OUT/b.log"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for sub in ("a", "b"):
    for f in glob.glob(sys.argv[1] + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            if "rocclr" in k: continue
            print(k)
            for c, v in sorted(cs.items()):
                print("    %-24s %14.0f  (x%d)" % (c, sum(v) / len(v), len(v)))
PY
