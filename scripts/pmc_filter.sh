#!/bin/bash
# PMC passes over the denoiser kernels (real frame, 1080p), one counter block per pass.
# Usage: scripts/pmc_filter.sh <tag> [library]   (outputs under gpurun_out/<tag>/)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/${1:-pmc_f}
if [ -n "$2" ]; then export PTK_LIB="$2"; fi
mkdir -p "$OUT"
pass() {
    n=$1; shift
    timeout -s KILL 90 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$n" -- python3 scripts/gpu_denoise_probe.py 1920 1080 4 real > "$OUT/$n.log" 2>&1 || exit 1
}
pass a TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY
pass b TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass c TCP_UTCL1_THRASHING_STALL TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS
pass d SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass e TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_RDREQ_LEVEL_sum
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for sub in "abcde":
    for f in glob.glob(sys.argv[1] + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in agg.items():
            if "filter" not in k and "features" not in k: continue
            print(k)
            for c, v in sorted(cs.items()):
                print("    %-44s %14.0f  (x%d)" % (c, sum(v) / len(v), len(v)))
PY
