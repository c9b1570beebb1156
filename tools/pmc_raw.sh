#!/bin/bash
# GPU box: one rocprofv3 --pmc pass (counters only, no tracing) over a python command; prints per-kernel means.
#   bash tools/pmc_raw.sh <tag> "<COUNTER COUNTER ...>" <script.py> [args ...]
set -o pipefail
TAG=$1; CTRS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CTRS --output-format csv -d "$OUT/pmc_$TAG" -o pmc -- python3 "$ROOT/$1" "${@:2}" > "$OUT/pmc_$TAG.log" 2>&1 \
    || { tail -5 "$OUT/pmc_$TAG.log"; exit 1; }
cd "$ROOT" && python3 - "$OUT/pmc_$TAG" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        if "mcn::" in row["Kernel_Name"]:
            acc[(row["Kernel_Name"].replace("void ", "").split("(")[0], int(row["Grid_Size"]))][row["Counter_Name"]].append(float(row["Counter_Value"]))
for (name, grid), ctr in sorted(acc.items()):
    print("%s grid %d (%d launches)" % (name[:50], grid, len(next(iter(ctr.values())))))
    for c, v in sorted(ctr.items()):
        print("    %-28s %16.1f" % (c, sum(v) / len(v)))
PY
