#!/bin/bash
# GPU box: the measurement set committed under profiles/ each round.  Usage: bash tools/profile_round.sh rNN
# (run through gpurun from the repo root; writes gpurun_out/prof_<tag>/ and the summaries to gpurun_out/profiles_<tag>/;
# copy those into profiles/ and commit).  Phase 1 collects the PMC counters (separate rocprofv3 --pmc passes, no
# tracing), phase 2 runs the driver's bench command under kernel tracing with the fresh summaries in place, so that
# the committed bench line carries roofline.traffic / roofline_valu derived from this very build.
set -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
DST=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT" "$DST"
cd /tmp && export TMPDIR=/tmp
KB="$ROOT/tools/kbench.py"
# 0. the vector-ALU issue rate the roofline_valu peak rests on (1 / 2 / 4 wavefronts per SIMD)
if [ -x "$ROOT/tools/microbench/valu_issue" ]; then
  timeout -k 10 120 "$ROOT/tools/microbench/valu_issue" > "$DST/${TAG}_valu_issue.txt" 2> "$OUT/valu_issue.err" || { tail -5 "$OUT/valu_issue.err"; exit 1; }
  cp "$DST/${TAG}_valu_issue.txt" "$ROOT/profiles/"
  echo "valu_issue done"
fi
# 1a. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes, counters only.  One run per workload:
#     o5 / g5 = single-step env kernels at 5 humans (ORCA / given velocities), o10 = 10 humans (config-5 shard and
#     2^18), r<T> = mcn_env_rollout with T steps per launch, sarl5 / sarl10 / sgan10 = the network kernels
pmc() {  # pmc <name> <kbench args...>
  local name=$1; shift
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d "$OUT/${name}_$C" -o pmc -- python3 "$KB" "$@" > "$OUT/${name}_$C.log" 2>&1 \
        || { tail -5 "$OUT/${name}_$C.log"; exit 1; }
  done
  echo "pmc $name done"
}
pmc o5  --sizes 4096,65536,1048576 --modes orca --iters 20
pmc g5  --sizes 4096,1048576 --modes given --no-hh --iters 20
pmc o10 --humans 10 --sizes 4096,262144 --modes orca --iters 20
for T in 20 100 1000; do pmc r$T --rollout $T --sizes 4096; done
for T in 50 200; do pmc r10_$T --rollout $T --humans 10 --sizes 4096; done      # env_step_loop_kernel<10, 0>
pmc sarl5  --sarl --humans 5
pmc sarl10 --sarl --humans 10
pmc sgan10 --sgan --humans 10 --sizes 4096 --iters 20
# 1b. instruction issue: SQ counters, one pass per workload
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"
sq() {  # sq <name> <kbench args...>
  local name=$1; shift
  rocprofv3 --pmc $SQ --output-format csv -d "$OUT/sq_$name" -o pmc -- python3 "$KB" "$@" > "$OUT/sq_$name.log" 2>&1 \
      || { tail -5 "$OUT/sq_$name.log"; exit 1; }
}
sq roll  --rollout 200 --sizes 4096
sq step  --sizes 4096,1048576 --modes orca --iters 20
sq given --sizes 1048576 --modes given --no-hh --iters 20
sq o10   --humans 10 --sizes 4096,262144 --modes orca --iters 20
sq roll10 --rollout 200 --humans 10 --sizes 4096
echo "pmc SQ done"
cd "$ROOT"
R() { echo "$OUT/$1_FETCH_SIZE:$OUT/$1_WRITE_SIZE:$2:$3:$4"; }
python3 tools/pmc_summary.py "$DST/${TAG}_pmc_env_step.json" --run "$(R o5 5 4096+65536+1048576 1)" \
    --run "$(R g5 5 4096+1048576 1)" --run "$(R o10 10 4096+262144 1)" \
    --note "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes (counter unit KB, FETCH_SIZE doubled per MI355X_MICROARCH.md, WRITE_SIZE as is); one kbench run per workload: ORCA humans x 5, given velocities x 5 (ModelCrowdSim.step: no human-human count), ORCA humans x 10."
python3 tools/pmc_summary.py "$DST/${TAG}_pmc_env_rollout.json" --run "$(R r20 5 4096 20)" --run "$(R r100 5 4096 100)" \
    --run "$(R r1000 5 4096 1000)" --run "$(R r10_50 10 4096 50)" --run "$(R r10_200 10 4096 200)"
python3 tools/pmc_summary.py "$DST/${TAG}_pmc_nets.json" --run "$(R sarl5 5 4096 1)" --run "$(R sarl10 10 4096 1)" \
    --run "$(R sgan10 10 4096 1)" --only sarl_,sgan_
python3 tools/pmc_sq_summary.py "$OUT/sq_roll,$OUT/sq_step,$OUT/sq_given,$OUT/sq_o10,$OUT/sq_roll10" "$DST/${TAG}_pmc_sq.json" \
    --spec "env_step_loop_kernel<10=loop:4096:10:200;env_rollout_quad_kernel<5=rollout:4096:5:200;env_step_quad_kernel<5=quad:4096:5:1;env_step_kernel<256, 5, 0, 0=fused:1048576:5:1;env_pair_kernel<5=pairwise:1048576:5:1;env_step_kernel<64, 10, 0, 0=fused:4096:10:1;env_step_kernel<256, 10, 0, 0=fused:262144:10:1"
# 1c. matrix-pipe counters of the network kernels (SARL look-ahead, SGAN step): one pass each
MF="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
cd /tmp
rocprofv3 --pmc $MF --output-format csv -d "$OUT/mfma_sarl" -o pmc -- python3 "$KB" --sarl --humans 5 \
    > "$OUT/mfma_sarl.log" 2>&1 || { tail -5 "$OUT/mfma_sarl.log"; exit 1; }
rocprofv3 --pmc $MF --output-format csv -d "$OUT/mfma_sgan" -o pmc -- python3 "$KB" --sgan --humans 10 --sizes 4096 --iters 20 \
    > "$OUT/mfma_sgan.log" 2>&1 || { tail -5 "$OUT/mfma_sgan.log"; exit 1; }
cd "$ROOT"
mkdir -p "$OUT/mfma_all" && cp -r "$OUT/mfma_sarl" "$OUT/mfma_sgan" "$OUT/mfma_all/"
python3 tools/pmc_mfma.py "$OUT/mfma_all" "$DST/${TAG}_pmc_mfma.json"
echo "pmc MFMA done"
# 2. the bench command itself under kernel tracing (same flags as the driver's N=1 run), PMC summaries in place
cp "$DST/${TAG}_pmc_env_step.json" "$DST/${TAG}_pmc_env_rollout.json" "$DST/${TAG}_pmc_sq.json" "$DST/${TAG}_pmc_nets.json" "$ROOT/profiles/"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 \
    > "$DST/${TAG}_bench_stdout.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$DST/${TAG}_bench_kernel_stats.csv"
# the headline kernel is launched with several lengths in one run: per-launch durations by class, from the trace
python3 "$ROOT/tools/trace_launches.py" "$(find "$OUT/stats" -name '*kernel_trace.csv' | head -1)" > "$DST/${TAG}_bench_rollout_launches.txt"
echo "bench + kernel stats done"
