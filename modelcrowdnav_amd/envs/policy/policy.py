"""Policy base class (reference: crowd_sim/envs/policy/policy.py:5-49)."""
import math

import numpy as np


class Policy(object):
    def __init__(self):
        self.trainable = False
        self.phase = None
        self.model = None
        self.device = None
        self.last_state = None
        self.time_step = None
        self.env = None     # set when the policy may query the env's dynamics

    def configure(self, config):
        raise NotImplementedError

    def set_phase(self, phase):
        self.phase = phase

    def set_device(self, device):
        self.device = device

    def set_env(self, env):
        self.env = env

    def get_model(self):
        return self.model

    def predict(self, state):
        raise NotImplementedError

    @staticmethod
    def reach_destination(state):
        s = state.self_state
        # numpy's 2-vector norm, as the reference (policy.py:46): sqrt(fma(x1, x1, x0 * x0)) on this image, the
        # formula the batched kernels and the oracle use -- not math.hypot, which can differ in the last bit
        return np.linalg.norm((s.py - s.gy, s.px - s.gx)) < s.radius
