#!/usr/bin/env python3
"""Per-kernel matrix-pipe summary of one rocprofv3 --pmc pass (csv) with the counters
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY.

    python tools/pmc_mfma.py <pmc_dir> [out.json]

Per launch means.  SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_16x16x4_f32), the SQ_WAVE_CYCLES /
SQ_WAIT_* / SQ_ACTIVE_INST_* family counts quad-cycles (MI355X_MICROARCH.md, cycle constants)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    acc = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if "mcn::" not in row["Kernel_Name"]:
                    continue
                key = (row["Kernel_Name"].replace("void ", "").split("(")[0], int(row["Grid_Size"]))
                acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
                if row["Counter_Name"] == "SQ_WAVES":
                    dur[key].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    out = []
    for key, ctr in sorted(acc.items()):
        m = {c: sum(v) / len(v) for c, v in ctr.items()}
        if not m.get("SQ_INSTS_MFMA"):
            continue
        us = sum(dur[key]) / len(dur[key]) / 1e3
        k = {"kernel": key[0], "grid_threads": key[1], "launches": len(ctr["SQ_WAVES"]), "avg_us_under_pmc": round(us, 1),
             "counters_per_launch": {c: round(v, 1) for c, v in sorted(m.items())},
             "mfma_per_wave": round(m["SQ_INSTS_MFMA"] / m["SQ_WAVES"], 1),
             # 1024 matrix pipes x the launch's duration at 2.4 GHz
             "pipe_busy_of_launch": round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * us * 2400.0), 4),
             "wave_wait_frac": round(m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"], 4),
             "wave_issue_stall_frac": round(m.get("SQ_WAIT_INST_ANY", 0) / m["SQ_WAVE_CYCLES"], 4),
             "wave_active_frac": round(m.get("SQ_ACTIVE_INST_ANY", 0) / m["SQ_WAVE_CYCLES"], 4)}
        out.append(k)
        print("%-40s grid %8d  %7.1f us  MFMA/wave %7.1f  busy cycles/MFMA %5.1f  pipe busy %5.1f %% of launch  "
              "wave: wait %4.1f %% issue-stall %4.1f %% active %4.1f %%" % (
                  key[0][:40], key[1], us, k["mfma_per_wave"], m["SQ_VALU_MFMA_BUSY_CYCLES"] / m["SQ_INSTS_MFMA"],
                  100 * k["pipe_busy_of_launch"], 100 * k["wave_wait_frac"], 100 * k["wave_issue_stall_frac"],
                  100 * k["wave_active_frac"]))
    if len(sys.argv) > 2:
        json.dump({"note": __doc__.split("\n\n")[0], "kernels": out}, open(sys.argv[2], "w"), indent=1)


if __name__ == "__main__":
    main()
