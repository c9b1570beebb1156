#!/bin/bash
# GPU box: the measurement set committed under profiles/ each round.  Usage: bash tools/profile_round.sh rNN
# (run through gpurun from the repo root; writes gpurun_out/prof_<tag>/ and the summaries to gpurun_out/profiles_<tag>/;
# copy those into profiles/ and commit).  Phase 1 collects the PMC counters (separate rocprofv3 --pmc passes, no
# tracing), phase 2 runs the driver's bench command under kernel tracing with the fresh summaries in place, so that
# the committed bench line carries roofline.traffic / roofline_valu derived from this very build.
set -o pipefail
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
DST=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT" "$DST"
cd /tmp && export TMPDIR=/tmp
KB="$ROOT/tools/kbench.py"
# 1a. HBM traffic of the env kernels: FETCH_SIZE and WRITE_SIZE in separate passes, counters only
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmc_$C" -o pmc -- python3 "$KB" \
      --sizes 4096,1048576 --modes orca --iters 20 > "$OUT/pmc_$C.log" 2>&1 || { tail -5 "$OUT/pmc_$C.log"; exit 1; }
  rocprofv3 --pmc $C --output-format csv -d "$OUT/pmcg_$C" -o pmc -- python3 "$KB" \
      --sizes 4096,1048576 --modes given --no-hh --iters 20 > "$OUT/pmcg_$C.log" 2>&1 || { tail -5 "$OUT/pmcg_$C.log"; exit 1; }
  for T in 20 100 1000; do
    rocprofv3 --pmc $C --output-format csv -d "$OUT/pmcr${T}_$C" -o pmc -- python3 "$KB" \
        --rollout $T --sizes 4096 > "$OUT/pmcr${T}_$C.log" 2>&1 || { tail -5 "$OUT/pmcr${T}_$C.log"; exit 1; }
  done
  echo "pmc $C done"
done
# 1b. instruction issue: SQ counters, one pass per workload
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY"
rocprofv3 --pmc $SQ --output-format csv -d "$OUT/sq_roll" -o pmc -- python3 "$KB" --rollout 200 --sizes 4096 \
    > "$OUT/sq_roll.log" 2>&1 || { tail -5 "$OUT/sq_roll.log"; exit 1; }
rocprofv3 --pmc $SQ --output-format csv -d "$OUT/sq_step" -o pmc -- python3 "$KB" --sizes 4096,1048576 \
    --modes orca --iters 20 > "$OUT/sq_step.log" 2>&1 || { tail -5 "$OUT/sq_step.log"; exit 1; }
rocprofv3 --pmc $SQ --output-format csv -d "$OUT/sq_given" -o pmc -- python3 "$KB" --sizes 1048576 \
    --modes given --no-hh --iters 20 > "$OUT/sq_given.log" 2>&1 || { tail -5 "$OUT/sq_given.log"; exit 1; }
echo "pmc SQ done"
cd "$ROOT"
python3 tools/pmc_summary.py "$OUT/pmc_FETCH_SIZE" "$OUT/pmc_WRITE_SIZE" "$OUT/step_orca.json" \
    --envs "env_step_kernel<256=1048576,env_step_kernel<64=4096,quad_kernel<5=4096"
python3 tools/pmc_summary.py "$OUT/pmcg_FETCH_SIZE" "$OUT/pmcg_WRITE_SIZE" "$OUT/step_given.json" \
    --envs "env_pair_kernel<5=1048576,env_step_kernel<64=4096"
for T in 20 100 1000; do
  python3 tools/pmc_summary.py "$OUT/pmcr${T}_FETCH_SIZE" "$OUT/pmcr${T}_WRITE_SIZE" "$OUT/roll_$T.json" --steps-per-launch $T --envs "quad_kernel<5=4096"
done
python3 - "$OUT" "$DST" "$TAG" <<'PY'
import json, sys
out, dst, tag = sys.argv[1:4]
a, b = json.load(open(out + "/step_orca.json")), json.load(open(out + "/step_given.json"))
json.dump({"note": a["note"] + " ORCA modes and given-velocity modes (ModelCrowdSim.step: no human-human count) in separate runs.",
           "kernels": a["kernels"] + b["kernels"]}, open("%s/%s_pmc_env_step.json" % (dst, tag), "w"), indent=1)
ks, note = [], None
for T in (20, 100, 1000):
    d = json.load(open("%s/roll_%d.json" % (out, T)))
    note = d["note"]
    ks += [k for k in d["kernels"] if "rollout" in k["kernel"]]
json.dump({"note": note + " One kbench --rollout T run per launch length T.", "kernels": ks},
          open("%s/%s_pmc_env_rollout.json" % (dst, tag), "w"), indent=1)
PY
python3 tools/pmc_sq_summary.py "$OUT/sq_roll,$OUT/sq_step,$OUT/sq_given" "$DST/${TAG}_pmc_sq.json" \
    --spec "env_rollout_quad_kernel<5=rollout:4096:5:200;env_step_quad_kernel<5=quad:4096:5:1;env_step_kernel<256, 5, 0, 0=fused:1048576:5:1;env_pair_kernel<5=pairwise:1048576:5:1"
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
cp "$DST/${TAG}_pmc_env_step.json" "$DST/${TAG}_pmc_env_rollout.json" "$DST/${TAG}_pmc_sq.json" "$ROOT/profiles/"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 \
    > "$DST/${TAG}_bench_stdout.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$DST/${TAG}_bench_kernel_stats.csv"
echo "bench + kernel stats done"
