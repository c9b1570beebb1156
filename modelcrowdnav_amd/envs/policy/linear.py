"""Straight-to-goal policy (reference: crowd_sim/envs/policy/linear.py:6-22)."""
import math

from .policy import Policy
from ..utils.action import ActionXY


class Linear(Policy):
    def __init__(self):
        super().__init__()
        self.trainable = False
        self.kinematics = "holonomic"
        self.multiagent_training = True

    def configure(self, config):
        return

    def predict(self, state):
        s = state.self_state
        heading = math.atan2(s.gy - s.py, s.gx - s.px)
        return ActionXY(math.cos(heading) * s.v_pref, math.sin(heading) * s.v_pref)
