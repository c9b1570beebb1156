#!/bin/bash
# GPU box: per-kernel times of one command under rocprofv3 kernel tracing.
#   bash tools/prof_kernels.sh <tag> <python script + args ...>      -> gpurun_out/r2/<tag>_kernel_stats.csv
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$TAG" -o t -- python3 "$ROOT/$1" "${@:2}" > "$OUT/prof_$TAG.log" 2>&1 \
    || { tail -5 "$OUT/prof_$TAG.log"; exit 1; }
find "$OUT/prof_$TAG" -name "*kernel_stats.csv" -exec cp {} "$OUT/${TAG}_kernel_stats.csv" \;
cut -d, -f1-8 "$OUT/${TAG}_kernel_stats.csv" | cut -c1-160 | head -12
