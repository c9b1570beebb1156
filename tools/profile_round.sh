#!/bin/bash
# GPU box: the measurement set committed under profiles/ each round.  Usage: bash tools/profile_round.sh rNN
# (run through gpurun from the repo root; writes gpurun_out/prof_<tag>/ and copies summaries to gpurun_out/profiles_<tag>/)
set -o pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
DST=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT" "$DST"
cd /tmp && export TMPDIR=/tmp
# 1. the bench command itself under kernel tracing (same flags as the driver's N=1 run)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 \
    > "$DST/${TAG}_bench_stdout.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$DST/${TAG}_bench_kernel_stats.csv"
echo "bench + kernel stats done"
# 2. HBM traffic of the env kernels: FETCH_SIZE and WRITE_SIZE in separate passes, counters only
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -o pmc -- python3 "$ROOT/tools/kbench.py" \
      --sizes 4096,1048576 --modes orca,given --iters 20 > "$OUT/pmc_$C.log" 2>&1 || { tail -5 "$OUT/pmc_$C.log"; exit 1; }
  for T in 20 100 1000; do
    rocprofv3 --pmc $C --output-format csv -d "$OUT/pmcr${T}_$C" -o pmc -- python3 "$ROOT/tools/kbench.py" \
        --rollout $T --sizes 4096 > "$OUT/pmcr${T}_$C.log" 2>&1 || { tail -5 "$OUT/pmcr${T}_$C.log"; exit 1; }
  done
  echo "pmc $C done"
done
# 3. instruction issue: SQ counters, one pass per workload
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"
rocprofv3 --pmc $SQ --output-format csv -d "$OUT/sq_roll" -o pmc -- python3 "$ROOT/tools/kbench.py" --rollout 200 --sizes 4096 \
    > "$OUT/sq_roll.log" 2>&1 || { tail -5 "$OUT/sq_roll.log"; exit 1; }
rocprofv3 --pmc $SQ --output-format csv -d "$OUT/sq_step" -o pmc -- python3 "$ROOT/tools/kbench.py" --sizes 4096,1048576 \
    --modes orca,given --iters 20 > "$OUT/sq_step.log" 2>&1 || { tail -5 "$OUT/sq_step.log"; exit 1; }
echo "pmc SQ done"
cd "$ROOT"
python3 tools/pmc_summary.py "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" "$DST/${TAG}_pmc_env_step.json" \
    --envs "env_step_kernel<256=1048576,env_step_kernel<64=4096,quad_kernel<5=4096"
for T in 20 100 1000; do
  python3 tools/pmc_summary.py "$OUT/pmcr${T}_FETCH_SIZE" "$OUT/pmcr${T}_WRITE_SIZE" "$OUT/roll_$T.json" --steps-per-launch $T --envs "quad_kernel<5=4096"
done
python3 - "$OUT" "$DST/${TAG}_pmc_env_rollout.json" <<'PY'
import json, sys
out, dst = sys.argv[1], sys.argv[2]
ks, note = [], None
for T in (20, 100, 1000):
    d = json.load(open("%s/roll_%d.json" % (out, T)))
    note = d["note"]
    ks += [k for k in d["kernels"] if "rollout" in k["kernel"]]
json.dump({"note": note + " One kbench --rollout T run per launch length T.", "kernels": ks}, open(dst, "w"), indent=1)
PY
python3 tools/pmc_sq_summary.py "$OUT/sq_roll,$OUT/sq_step" "$DST/${TAG}_pmc_sq.json" \
    --spec "env_rollout_quad_kernel<5=rollout:4096:5:200;env_step_quad_kernel<5=quad:4096:5:1;env_step_kernel<256, 5, 0, 0=fused:1048576:5:1;env_step_kernel<256, 0, 0, 2=pairwise:1048576:5:1"
