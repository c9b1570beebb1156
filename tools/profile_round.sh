#!/bin/bash
# GPU box: the measurement set committed under profiles/ each round.  Usage: bash tools/profile_round.sh rNN
# (run through gpurun from the repo root; writes gpurun_out/prof_<tag>/ and copies summaries to gpurun_out/profiles_<tag>/)
set -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
DST=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT" "$DST"
cd /tmp && export TMPDIR=/tmp
# 1. the bench command itself under kernel tracing (same flags as the driver's N=1 run)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 "$ROOT/bench.py" \
    > "$DST/${TAG}_bench_stdout.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$DST/${TAG}_bench_kernel_stats.csv"
echo "bench + kernel stats done"
# 2. HBM traffic of the env kernels: FETCH_SIZE and WRITE_SIZE in separate passes, counters only
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -o pmc -- python3 "$ROOT/tools/kbench.py" \
      --sizes 4096,1048576 --modes orca,given --iters 20 > "$OUT/pmc_$C.log" 2>&1 || { tail -5 "$OUT/pmc_$C.log"; exit 1; }
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmcr_$C" -o pmc -- python3 "$ROOT/tools/kbench.py" \
      --rollout 1000 --sizes 4096 > "$OUT/pmcr_$C.log" 2>&1 || { tail -5 "$OUT/pmcr_$C.log"; exit 1; }
  echo "pmc $C done"
done
cd "$ROOT"
python3 tools/pmc_summary.py "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" "$DST/${TAG}_pmc_env_step.json" \
    --envs "env_step_kernel<256=1048576,env_step_kernel<64=4096,quad_kernel<5=4096"
python3 tools/pmc_summary.py "$OUT/pmcr_FETCH_SIZE" "$OUT/pmcr_WRITE_SIZE" "$DST/${TAG}_pmc_env_rollout.json" --steps-per-launch 1000 --envs "quad_kernel<5=4096"
