#!/usr/bin/env python3
"""Per-launch durations of one kernel from a rocprofv3 --kernel-trace CSV, grouped by launch length.

    python tools/trace_launches.py <kernel_trace.csv> [kernel substring] > profiles/rNN_bench_rollout_launches.txt

bench.py's headline kernel (mcn::env_rollout_quad_kernel) is launched with several lengths in one run -- the
1000-step launches of the timed region and of the probe that sizes it, 20-step probe / `short_launch` passes, a few
warm-up launches -- so the `--stats` average over all of them says nothing.  This lists every launch in dispatch order
and averages each duration class (a launch's time is proportional to its steps), so that the class of the timed
region can be compared with `roofline.avg_launch_us` of the same run.
"""
import csv
import sys


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else "env_rollout_quad_kernel"
    rows = []
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            if want in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
                             int(r["Grid_Size_X"]), r["Kernel_Name"].replace("void ", "").split("(")[0]))
    rows.sort()
    if not rows:
        print("no launches of", want)
        return
    print("kernel: %s, %d launches; duration in us (grid threads), dispatch order:" % (rows[0][3], len(rows)))
    print(" ".join("%.1f(%d)" % (r[1], r[2]) for r in rows))
    # duration classes per grid: launches within a factor 1.5 of each other
    for grid in sorted(set(r[2] for r in rows)):
        classes = []
        for d in sorted(r[1] for r in rows if r[2] == grid):
            if classes and d <= 1.5 * classes[-1][0]:
                classes[-1].append(d)
            else:
                classes.append([d])
        for c in classes:
            print("grid %7d  class %8.1f - %8.1f us: %3d launches, mean %9.2f us" % (grid, c[0], c[-1], len(c), sum(c) / len(c)))
    big = max(r[2] for r in rows)
    longest = [r[1] for r in rows if r[2] == big]
    top = [d for d in longest if d >= 0.5 * max(longest)]
    print("the timed region's launches are the LAST ones of the longest class (its first two are the probe that sizes "
          "the region): mean of all but the first two = %.2f us over %d launches" % (
              sum(top[2:]) / max(1, len(top) - 2), max(0, len(top) - 2)))


if __name__ == "__main__":
    main()
