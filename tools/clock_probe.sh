#!/bin/bash
# GPU box: sample the shader clock while a kernel loop runs (is the matrix pipe throttled below its 2.4 GHz peak?).
#   bash tools/clock_probe.sh <python script + args ...>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$ROOT/gpurun_out/r2"
python3 "$ROOT/$1" "${@:2}" > "$ROOT/gpurun_out/r2/clock_probe_run.log" 2>&1 &
PID=$!
sleep 12
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power (W)\|Socket" | head -4
  sleep 1
done
wait $PID
tail -3 "$ROOT/gpurun_out/r2/clock_probe_run.log"
