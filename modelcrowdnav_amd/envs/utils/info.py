"""Step outcome markers (reference: crowd_sim/envs/utils/info.py:1-38).

Each class carries the integer `code` the HIP kernels emit (include/mcn.h MCN_INFO_*), so a
batch of uint8 codes converts to the reference's info objects with `from_code`.
"""


class _Outcome(object):
    code = -1
    label = ""

    def __str__(self):
        return self.label


class Nothing(_Outcome):
    code, label = 0, ""


class Danger(_Outcome):
    code, label = 1, "Too close"

    def __init__(self, min_dist):
        self.min_dist = min_dist


class ReachGoal(_Outcome):
    code, label = 2, "Reaching goal"


class Collision(_Outcome):
    code, label = 3, "Collision"


class Timeout(_Outcome):
    code, label = 4, "Timeout"


def from_code(code, dmin=None):
    code = int(code)
    if code == Danger.code:
        return Danger(dmin)
    return (Nothing, None, ReachGoal, Collision, Timeout)[code]()
