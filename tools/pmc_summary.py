#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --output-format csv) over the same command into the
per-kernel HBM-traffic summary committed under profiles/.

    python tools/pmc_summary.py <fetch_dir> <write_dir> <out.json> [--steps-per-launch S]

Counter unit: KB per dispatch.  Per /opt/skills/guides/MI355X_MICROARCH.md ("HBM"), FETCH_SIZE on gfx950 reports
half of the bytes of wide coalesced loads, so reads are doubled; WRITE_SIZE is taken as is.  Only mcn:: kernels are
kept; launches are grouped by (kernel name, grid size) and averaged.
"""
import argparse
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def collect(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter or "mcn::" not in row["Kernel_Name"]:
                    continue
                name = row["Kernel_Name"].replace("void ", "").split("(")[0]
                acc[(name, int(row["Grid_Size"]), int(row["Workgroup_Size"]))].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def envs_of(name, grid, wg):
    """Batch size from the launch geometry of the env kernels (see the launchers in csrc/)."""
    import re
    m = re.search(r"env_(?:step|rollout)_quad_kernel<(\d+), (\d+)", name)
    if m:                                   # one wavefront (two when split) per 64 // (4 N) envs
        nt = int(m.group(1))
        return None, grid // wg, 64 // (4 * nt)
    return None, None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir"); ap.add_argument("write_dir"); ap.add_argument("out")
    ap.add_argument("--humans", type=int, default=5)
    ap.add_argument("--envs", type=str, default="", help="kernel-substring=envs[,..] overrides, e.g. 'env_step_kernel<256=1048576'")
    ap.add_argument("--steps-per-launch", type=int, default=1, help="steps one env_rollout launch advances")
    ap.add_argument("--note", default="")
    a = ap.parse_args()
    import bench
    fetch, n_f = collect(a.fetch_dir, "FETCH_SIZE")
    write, _ = collect(a.write_dir, "WRITE_SIZE")
    over = dict(x.split("=") for x in a.envs.split(",") if x)
    kernels = []
    for key in sorted(fetch, key=lambda k: (k[1], k[0])):
        name, grid, wg = key
        if key not in write or "env_" not in name:
            continue
        E = None
        for sub, val in over.items():
            if sub in name:
                E = int(val)
        _, blocks, per = envs_of(name, grid, wg)
        if E is None and blocks is not None:
            E = blocks * per                          # upper bound; exact when E divides evenly
        if E is None:
            E = grid // a.humans                      # lane-per-human kernels: about one lane per human
        given = ", 2, " in name                       # MODE template argument = MCN_HUMANS_GIVEN
        S = a.steps_per_launch if "rollout" in name else 1
        alg = (bench.pairwise_bytes_per_env_step(a.humans) if given else bench.algorithmic_bytes_per_env_step(a.humans)) * E * S
        rd, wr = 2.0 * fetch[key] * 1024.0, write[key] * 1024.0
        kernels.append({
            "kernel": name, "grid_threads": grid, "workgroup": wg, "launches_averaged": n_f[key], "envs": E,
            "steps_per_launch": S,
            "what": ("pairwise-only (given velocities)" if given else "fused ORCA step") + (", %d steps per launch" % S if S > 1 else ""),
            "FETCH_SIZE_KB": round(fetch[key], 1), "WRITE_SIZE_KB": round(write[key], 1),
            "hbm_read_bytes": int(rd), "hbm_write_bytes": int(wr), "traffic_bytes_per_launch": int(rd + wr),
            "algorithmic_bytes_per_launch": int(alg), "traffic_over_algorithmic": round((rd + wr) / alg, 4)})
    json.dump({"note": a.note or "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, --output-format csv; "
               "counter unit KB, averaged over the launches of each (kernel, grid); FETCH_SIZE doubled per "
               "MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced loads), WRITE_SIZE as is.",
               "kernels": kernels}, open(a.out, "w"), indent=1)
    for k in kernels:
        print("%-60s E=%8d S=%3d traffic/alg = %.3f" % (k["kernel"][:60], k["envs"], k["steps_per_launch"], k["traffic_over_algorithmic"]))


if __name__ == "__main__":
    main()
