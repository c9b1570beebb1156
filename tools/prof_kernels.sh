#!/bin/bash
# GPU box: per-kernel times (rocprofv3 --kernel-trace --stats) of one python command; prints the mcn:: rows.
#   bash tools/prof_kernels.sh <tag> <script.py> [args ...]
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3/kprof_$TAG
mkdir -p "$OUT"
(cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- python3 "$ROOT/$1" "${@:2}" > "$OUT.log" 2>&1) || { tail -5 "$OUT.log"; exit 1; }
grep -v amdgpu.ids "$OUT.log" | tail -4
find "$OUT" -name "*kernel_stats.csv" -exec grep "mcn::" {} \; | cut -d, -f1-4,6-7 | cut -c1-150
