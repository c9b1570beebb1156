"""Host-side agent record (reference: crowd_sim/envs/utils/agent.py:10-138).

In this build an Agent is a *mirror*: authoritative state lives in HBM inside VecCrowdSim and
is advanced by env_step.hip.  The E = 1 gym view (CrowdSim) refreshes these records after
every step so that reference callers reading `env.humans[i].px` / `robot.get_position()` keep
working.  The kinematics helpers below are therefore only used by host-side callers
(policies' one-step propagate, drivers), never by the env step itself.
"""
import logging
import math

from .action import ActionXY, ActionRot
from .state import ObservableState, FullState
from ..policy.policy_factory import policy_factory

_TWO_PI = 2 * math.pi


class Agent(object):
    def __init__(self, config, section):
        self.visible = config.getboolean(section, "visible")
        self.v_pref = config.getfloat(section, "v_pref")
        self.radius = config.getfloat(section, "radius")
        self.policy = policy_factory[config.get(section, "policy")]()
        self.sensor = config.get(section, "sensor")
        self.kinematics = None if self.policy is None else self.policy.kinematics
        self.px = self.py = self.gx = self.gy = self.vx = self.vy = self.theta = None
        self.time_step = None

    # -- configuration ---------------------------------------------------------------------
    def print_info(self):
        logging.info("Agent is %s and has %s kinematic constraint",
                     "visible" if self.visible else "invisible", self.kinematics)

    def set_policy(self, policy):
        self.policy = policy
        self.kinematics = policy.kinematics

    def sample_random_attributes(self):
        """agent.py:39-45 -- two draws from the global numpy stream, v_pref first."""
        import numpy as np
        self.v_pref = np.random.uniform(0.5, 1.5)
        self.radius = np.random.uniform(0.3, 0.5)

    def set(self, px, py, gx, gy, vx, vy, theta, radius=None, v_pref=None):
        self.px, self.py, self.gx, self.gy = px, py, gx, gy
        self.vx, self.vy, self.theta = vx, vy, theta
        if radius is not None:
            self.radius = radius
        if v_pref is not None:
            self.v_pref = v_pref

    # -- views -----------------------------------------------------------------------------
    def get_observable_state(self):
        return ObservableState(self.px, self.py, self.vx, self.vy, self.radius)

    def get_full_state(self):
        return FullState(self.px, self.py, self.vx, self.vy, self.radius, self.gx, self.gy, self.v_pref, self.theta)

    def get_position(self):
        return self.px, self.py

    def set_position(self, position):
        self.px, self.py = position[0], position[1]

    def get_goal_position(self):
        return self.gx, self.gy

    def get_velocity(self):
        return self.vx, self.vy

    def set_velocity(self, velocity):
        self.vx, self.vy = velocity[0], velocity[1]

    # -- kinematics ------------------------------------------------------------------------
    def check_validity(self, action):
        want = ActionXY if self.kinematics == "holonomic" else ActionRot
        assert isinstance(action, want)

    def _heading_velocity(self, action, theta):
        return action.v * math.cos(theta), action.v * math.sin(theta)

    def compute_position(self, action, delta_t):
        self.check_validity(action)
        if self.kinematics == "holonomic":
            return self.px + action.vx * delta_t, self.py + action.vy * delta_t
        theta = self.theta + action.r
        return self.px + math.cos(theta) * action.v * delta_t, self.py + math.sin(theta) * action.v * delta_t

    def get_next_observable_state(self, action):
        npx, npy = self.compute_position(action, self.time_step)
        if self.kinematics == "holonomic":
            nvx, nvy = action.vx, action.vy
        else:
            nvx, nvy = self._heading_velocity(action, self.theta + action.r)
        return ObservableState(npx, npy, nvx, nvy, self.radius)

    def step(self, action):
        self.px, self.py = self.compute_position(action, self.time_step)
        if self.kinematics == "holonomic":
            self.vx, self.vy = action.vx, action.vy
        else:
            self.theta = (self.theta + action.r) % _TWO_PI
            self.vx, self.vy = self._heading_velocity(action, self.theta)

    def reached_destination(self):
        import numpy as np
        gap = np.array(self.get_position()) - np.array(self.get_goal_position())
        return np.linalg.norm(gap) < self.radius

    def act(self, ob):
        raise NotImplementedError
