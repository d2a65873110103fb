#!/bin/bash
# Runs on the GPU box: kernel-trace statistics of the denoiser at 3840x2160 on a rendered frame
# (scripts/gpu_denoise_probe.py), where its stencils stream from HBM rather than from the
# Infinity Cache.  Usage: scripts/collect_denoise_4k.sh <tag>   (outputs under gpurun_out/<tag>/)
set -e
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-dn4k}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- \
    python3 scripts/gpu_denoise_probe.py 3840 2160 12 real > "$OUT/probe.txt"
cat "$OUT/probe.txt"
