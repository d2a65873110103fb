#!/bin/bash
# Runs on the GPU box: kernel-trace statistics of the default bench command and the
# PMC passes (one counter block per pass) behind bench.py's roofline object.
# Usage: scripts/collect_profiles.sh <tag>      (outputs under gpurun_out/<tag>/)
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
SHORT="--steps 2 --warmup 1 --spp-per-step 200 --no-cpu-baseline --no-denoise-4k"

timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- \
    python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-denoise-4k > "$OUT/bench_under_rocprof.json"
echo "kernel trace done"
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES \
    --output-format csv -d "$OUT/pmc_sq" -- python3 bench.py $SHORT > "$OUT/pmc_sq.json"
echo "pmc sq done"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- \
    python3 bench.py $SHORT > "$OUT/pmc_fetch.json"
echo "pmc fetch done"
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- \
    python3 bench.py $SHORT > "$OUT/pmc_write.json"
echo "pmc write done"
python3 scripts/summarize_profiles.py "$OUT" > "$OUT/summary.json"
cat "$OUT/summary.json"
