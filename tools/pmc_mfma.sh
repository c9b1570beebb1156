#!/bin/bash
# GPU box: one SQ counter pass (no tracing) over a python command, summarised per kernel.
#   bash tools/pmc_mfma.sh <tag> <script.py> [args ...]   -> gpurun_out/r2/pmc_<tag>/, gpurun_out/r2/pmc_<tag>.json
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
    --output-format csv -d "$OUT/pmc_$TAG" -o pmc -- python3 "$ROOT/$1" "${@:2}" > "$OUT/pmc_$TAG.log" 2>&1 \
    || { tail -5 "$OUT/pmc_$TAG.log"; exit 1; }
cd "$ROOT" && python3 tools/pmc_mfma.py "$OUT/pmc_$TAG" "$OUT/pmc_$TAG.json"
