#!/bin/bash
# GPU box: why does the streaming given-velocity kernel (env_pair_kernel) reach a lower share of HBM at 2^22 envs than at
# 2^20?    Address-translation and L2 / memory-side request counters of the SAME launch shape at both
# sizes, counters only (no tracing), one rocprofv3 pass per counter group:
#   bash tools/pmc_pair.sh r04     -> gpurun_out/prof_pair_<tag>/, summary gpurun_out/profiles_<tag>/<tag>_pmc_pair.json
set -o pipefail
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_pair_$TAG
DST=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT" "$DST"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters.txt" 2>&1 || true
grep -o -i "\b\(TCP_UTCL1[A-Z0-9_]*\|TCC_[A-Z0-9_]*\|TCP_[A-Z0-9_]*STALL[A-Z0-9_]*\|GRBM_GUI_ACTIVE\|TCP_TCC_[A-Z0-9_]*\)\b" "$OUT/counters.txt" | sort -u > "$OUT/names.txt"
KB="$ROOT/tools/kbench.py"
run() {  # run <group name> <counters...>   (two or three counters per pass: five TCC_*_sum in one pass abort rocprofv3)
  local g=$1; shift
  for E in 1048576 4194304; do
    echo "$(date +%T) group $g at $E: $*" >> "$OUT/progress.txt"
    timeout -k 10 200 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/${g}_$E" -o pmc -- python3 "$KB" --sizes $E --modes given --no-hh --iters 20 \
        > "$OUT/${g}_$E.log" 2>&1 || { echo "group $g at $E failed" >> "$OUT/progress.txt"; tail -3 "$OUT/${g}_$E.log" >> "$OUT/progress.txt"; }
  done
  echo "group $g done"
}
have() { grep -qx "$1" "$OUT/names.txt"; }
pick() { local o=""; for c in "$@"; do have $c && o="$o $c"; done; echo $o; }
G1=$(pick TCP_UTCL1_REQUEST TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_PERMISSION_MISS)
G2=$(pick TCC_HIT_sum TCC_MISS_sum)
G3=$(pick TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum)
G4=$(pick TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum)
G5=$(pick TCC_TAG_STALL_sum TCC_EA0_WRREQ_STALL_sum)
G6=$(pick TCC_BUSY_sum TCC_CYCLE_sum)
G7=$(pick TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum)
G8=$(pick TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_SRC_FIFO_FULL_sum)
[ -n "$G1" ] && [ ! -d "$OUT/utcl1_4194304" ] && run utcl1 $G1
[ -n "$G2" ] && run tcc_hit $G2
[ -n "$G3" ] && run tcc_ea $G3
[ -n "$G4" ] && run tcc_credit $G4
[ -n "$G5" ] && run tcc_stall $G5
[ -n "$G6" ] && run tcc_busy $G6
[ -n "$G7" ] && run tcc_level $G7
[ -n "$G8" ] && run tcc_fifo $G8
cd "$ROOT"
python3 - "$OUT" "$DST/${TAG}_pmc_pair.json" <<'PY'
import csv, glob, json, os, sys
out, dst = sys.argv[1], sys.argv[2]
res = {}
for d in sorted(glob.glob(os.path.join(out, "*_[0-9]*"))):
    if not os.path.isdir(d):
        continue
    grp, E = os.path.basename(d).rsplit("_", 1)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        acc = {}
        for row in csv.DictReader(open(f)):
            if "env_pair_kernel" not in row.get("Kernel_Name", ""):
                continue
            k = row["Counter_Name"]
            acc.setdefault(k, []).append(float(row["Counter_Value"]))
        for k, v in acc.items():
            res.setdefault(E, {})[k] = {"per_launch": sum(v) / len(v), "launches": len(v)}
json.dump({"kernel": "mcn::env_pair_kernel<5, true>", "workload": "kbench --modes given --no-hh, 20 launches per size",
           "by_envs": res}, open(dst, "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY
