#!/bin/bash
# GPU box: the whole `-m gpu` suite once per alternative dispatch default (the MCN_* variables give mcn_tuning's initial
# values), so that every caller-level test also runs on the decompositions the dispatcher does not pick by itself.
#   bash tools/tuning_matrix.sh [outdir]     -> <outdir>/matrix_<n>.log, one summary line each on stdout
# A run that hits its time limit stops the script (no further GPU step after a killed one).
OUT=${1:-gpurun_out/matrix}
mkdir -p "$OUT"
n=0
for ENVS in "MCN_LP3_DEFER=1" "MCN_STEP_BLOCK=256 MCN_QUAD_MAX_ENVS=0" "MCN_FORCE_GENERIC=1" "MCN_PAIR_STREAM=2 MCN_ROLLOUT_FUSED=0" "MCN_SARL_X3=0"; do
  n=$((n + 1))
  echo "== $ENVS" > "$OUT/matrix_$n.log"
  env $ENVS timeout -k 10 900 python -m pytest tests/ -m gpu -q >> "$OUT/matrix_$n.log" 2>&1
  rc=$?
  echo "$ENVS: rc=$rc $(tail -n 1 "$OUT/matrix_$n.log")"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out: stopping"; exit 1; fi
done
