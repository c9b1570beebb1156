#!/bin/bash
# GPU box: per-kernel times (rocprofv3 kernel trace) of one python command against each A/B library.
#   bash tools/ab_prof.sh "<variant> ..." <kernel substring> <script.py> [args ...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
VARS=$1; PAT=$2; shift 2
mkdir -p "$ROOT/gpurun_out/r2"
for v in $VARS; do
  export MCN_HIP_LIB=$ROOT/modelcrowdnav_amd/csrc/build_ab/$v/libmcn_hip.so
  echo "== $v"
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/gpurun_out/r2/abprof_$v" -o t -- python3 "$ROOT/$1" "${@:2}" > "$ROOT/gpurun_out/r2/abprof_$v.log" 2>&1) || { tail -3 "$ROOT/gpurun_out/r2/abprof_$v.log"; exit 1; }
  find "$ROOT/gpurun_out/r2/abprof_$v" -name "*kernel_stats.csv" -exec grep "$PAT" {} \; | cut -d, -f1-4 | cut -c1-120
done
