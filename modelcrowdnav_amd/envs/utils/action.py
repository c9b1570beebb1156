"""Robot/human action tuples (reference: crowd_sim/envs/utils/action.py:1-4)."""
import collections

ActionXY = collections.namedtuple("ActionXY", "vx vy")     # holonomic: velocity components
ActionRot = collections.namedtuple("ActionRot", "v r")     # unicycle: speed, heading change
