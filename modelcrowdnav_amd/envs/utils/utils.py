"""Geometry helper kept for API compatibility (reference: crowd_sim/envs/utils/utils.py:4-26).

The env never calls this on the host: the swept-circle test runs inside env_step.hip
(p2s_origin).  Host callers (plot/debug code) get the same arithmetic here.
"""
import math


def point_to_segment_dist(x1, y1, x2, y2, x3, y3):
    sx, sy = x2 - x1, y2 - y1
    if sx == 0 and sy == 0:
        return math.hypot(x3 - x1, y3 - y1)
    u = ((x3 - x1) * sx + (y3 - y1) * sy) / (sx * sx + sy * sy)
    u = min(1.0, max(0.0, u))
    return math.hypot(x1 + u * sx - x3, y1 + u * sy - y3)
