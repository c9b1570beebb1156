#!/usr/bin/env python3
"""Turn one rocprofv3 --pmc pass of SQ counters (--output-format csv) into the per-kernel instruction-issue summary
committed under profiles/ (read back by bench.py's roofline_valu).

    python tools/pmc_sq_summary.py <pmc_dir>[,<pmc_dir>...] <out.json> --spec "substr=kind:envs:humans:steps[,...]"

--spec names the launches to keep: a kernel whose name contains `substr` AND whose grid matches is summarised as
`kind` ("rollout" | "quad" | "fused" | "pairwise" | ...) run on `envs` envs of `humans` humans for `steps` steps
per launch.  Counters are per-launch means.  SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles
(/opt/skills/guides/MI355X_MICROARCH.md, cycle constants table).
"""
import argparse
import csv
import glob
import json
import os
from collections import defaultdict


def collect(dirs):
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(f, newline="") as fh:
                for row in csv.DictReader(fh):
                    if "mcn::" not in row["Kernel_Name"]:
                        continue
                    name = row["Kernel_Name"].replace("void ", "").split("(")[0]
                    acc[(name, int(row["Grid_Size"]), int(row["Workgroup_Size"]))][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs"); ap.add_argument("out"); ap.add_argument("--spec", required=True)
    ap.add_argument("--note", default="")
    a = ap.parse_args()
    acc = collect(a.dirs.split(","))
    specs = []
    for item in a.spec.split(";"):
        if not item:
            continue
        sub, rest = item.split("=")
        kind, envs, humans, steps = rest.split(":")
        specs.append((sub, kind, int(envs), int(humans), int(steps)))
    kernels = []
    for (name, grid, wg), ctr in sorted(acc.items()):
        for sub, kind, envs, humans, steps in specs:
            if sub not in name:
                continue
            m = {c: sum(v) / len(v) for c, v in ctr.items()}
            waves = m.get("SQ_WAVES", 0.0)
            if not waves:
                continue
            # the spec's env count must be what this grid covers (several batch sizes share one kernel name)
            lanes_needed = {"rollout": 4 * humans * envs, "quad": 4 * humans * envs}.get(kind, humans * envs)
            if not (0.45 * grid <= lanes_needed <= 2.2 * grid):
                continue
            es = envs * steps
            insts = sum(m.get(c, 0.0) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM",
                                                "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_FLAT"))
            k = {"kernel": name, "kind": kind, "grid_threads": grid, "workgroup": wg, "envs": envs, "humans": humans,
                 "steps_per_launch": steps, "launches_averaged": len(ctr["SQ_WAVES"]),
                 "counters_per_launch": {c: round(v, 1) for c, v in sorted(m.items())},
                 "waves": waves,
                 "valu_per_env_step": m.get("SQ_INSTS_VALU", 0.0) / es,
                 "insts_per_env_step": insts / es,
                 "valu_per_wave_step": m.get("SQ_INSTS_VALU", 0.0) / waves / steps,
                 "insts_per_wave_step": insts / waves / steps}
            if m.get("SQ_WAVE_CYCLES"):
                k["wave_cycles_per_wave_step"] = 4.0 * m["SQ_WAVE_CYCLES"] / waves / steps
                if "SQ_WAIT_ANY" in m:
                    k["wait_frac"] = round(m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"], 4)
            kernels.append(k)
            print("%-58s %-8s E=%8d T=%4d  VALU/env-step %8.1f  insts/wave-step %7.1f  wait %s" % (
                name[:58], kind, envs, steps, k["valu_per_env_step"], k["insts_per_wave_step"], k.get("wait_frac")))
    json.dump({"note": a.note or "rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES "
               "SQ_ACTIVE_INST_VALU SQ_WAIT_ANY, one pass, --output-format csv; per-launch means.",
               "kernels": kernels}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
