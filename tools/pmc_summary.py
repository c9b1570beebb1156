#!/usr/bin/env python3
"""Turn pairs of rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; --output-format csv) over the same command into the
per-kernel HBM-traffic summary committed under profiles/.

    python tools/pmc_summary.py <out.json> --run <fetch_dir>:<write_dir>:<humans>:<E1+E2+...>:<steps_per_launch> [--run ...]

One --run per profiled command: what the command stepped (humans per env, the batch sizes it ran, steps per launch for
mcn_env_rollout).  Counter unit: KB per dispatch.  Per /opt/skills/guides/MI355X_MICROARCH.md ("HBM"), FETCH_SIZE on
gfx950 reports half of the bytes of wide coalesced loads, so reads are doubled; WRITE_SIZE is taken as is.  Only mcn::
kernels are kept; launches are grouped by (kernel name, grid size) and averaged.

Every entry says which kernel FAMILY it is (env_pair_kernel / env_step_kernel / env_step_quad_kernel /
env_rollout_quad_kernel / env_step_loop_kernel / sarl_value_kernel / sgan_*_kernel), which human-policy MODE that instantiation computes
(parsed from the template arguments, never guessed from a substring) and on how many envs of how many humans it ran
(the run's batch size whose lane count matches the launch's grid); bench.py matches on exactly those.
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
import bench  # noqa: E402  (classify_kernel, the byte accounting)


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


def match_envs(grid, humans, lanes_per_human, sizes):
    """The run's batch size this launch covered: grid = E x humans x lanes-per-human, padded to whole wavefronts
    (at most 64 / 60 for the layouts in csrc/) and workgroups."""
    for E in sizes:
        need = E * humans * lanes_per_human
        if need <= grid <= 1.4 * need + 1024:
            return E
    return None


def summarise(run, algorithmic):
    fdir, wdir, humans, sizes, spl = run
    fetch, n_f = collect(fdir, "FETCH_SIZE")
    write, _ = collect(wdir, "WRITE_SIZE")
    out = []
    for key in sorted(fetch, key=lambda k: (k[1], k[0])):
        name, grid, wg = key
        if key not in write:
            continue
        c = bench.classify_kernel(name, humans)
        if c is None:
            continue
        fam, n, mode, lph = c
        if lph is None:                      # network kernels: one batch size per run, grid is not a lane-per-human count
            E = sizes[0] if len(sizes) == 1 else None
        else:
            E = match_envs(grid, n, lph, sizes)
        if E is None:
            continue
        S = spl if fam in ("env_rollout_quad_kernel", "env_step_loop_kernel") else 1
        rd, wr = 2.0 * fetch[key] * 1024.0, write[key] * 1024.0
        e = {"kernel": name, "family": fam, "mode": mode, "humans": n, "envs": E, "steps_per_launch": S,
             "grid_threads": grid, "workgroup": wg, "launches_averaged": n_f[key],
             "FETCH_SIZE_KB": round(fetch[key], 1), "WRITE_SIZE_KB": round(write[key], 1),
             "hbm_read_bytes": int(rd), "hbm_write_bytes": int(wr), "traffic_bytes_per_launch": int(rd + wr)}
        if mode is not None:
            alg = algorithmic(n, mode == "given") * E * S
            e["what"] = {"given": "pairwise-only (given velocities, 566-B accounting at 5 humans)",
                         "orca": "fused ORCA step", "linear": "linear humans"}[mode] + (", %d steps per launch" % S if S > 1 else "")
            e["algorithmic_bytes_per_launch"] = int(alg)
            e["traffic_over_algorithmic"] = round((rd + wr) / alg, 4)
        out.append(e)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--run", action="append", required=True,
                    help="fetch_dir:write_dir:humans:E1+E2+..:steps_per_launch")
    ap.add_argument("--note", default="")
    ap.add_argument("--only", default="", help="comma list of kernel-name substrings to keep (default: all mcn:: kernels)")
    a = ap.parse_args()
    alg = lambda n, given: bench.pairwise_bytes_per_env_step(n) if given else bench.algorithmic_bytes_per_env_step(n)
    kernels = []
    for r in a.run:
        fdir, wdir, humans, sizes, spl = r.split(":")
        kernels += summarise((fdir, wdir, int(humans), [int(x) for x in sizes.split("+")], int(spl)), alg)
    if a.only:
        kernels = [k for k in kernels if any(sub in k["kernel"] for sub in a.only.split(","))]
    json.dump({"note": a.note or "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, --output-format csv; "
               "counter unit KB, averaged over the launches of each (kernel, grid); FETCH_SIZE doubled per "
               "MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced loads), WRITE_SIZE as is.",
               "kernels": kernels}, open(a.out, "w"), indent=1)
    for k in kernels:
        print("%-58s %-6s N=%2d E=%8d S=%4d  %10.1f KB  traffic/alg = %s" % (
            k["kernel"][:58], k["mode"], k["humans"], k["envs"], k["steps_per_launch"],
            k["traffic_bytes_per_launch"] / 1024.0, k.get("traffic_over_algorithmic")))


if __name__ == "__main__":
    main()
