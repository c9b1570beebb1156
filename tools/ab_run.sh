#!/bin/bash
# GPU box: run one python tool against each A/B library built by tools/ab_build.sh.
#   bash tools/ab_run.sh "<variant> <variant> ..." <script.py> [args ...]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
VARS=$1; shift
for v in $VARS; do
  echo "== $v"
  W=$(echo $v | sed "s/.*w\([0-9]*\)$/\1/")
  MCN_POOL_WAVES=${W:-16} MCN_HIP_LIB=$ROOT/modelcrowdnav_amd/csrc/build_ab/$v/libmcn_hip.so timeout -k 10 120 python3 "$ROOT/$1" "${@:2}" 2>&1 | grep -v amdgpu.ids | tail -6 || exit 1
done
