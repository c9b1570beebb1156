"""Shared pieces of the value-based robot policies (reference: crowd_nav/policy/cadrl.py).

  mlp()                 cadrl.py:11-19   nn.Sequential of Linear/ReLU with the reference's key names
  build_action_space()  cadrl.py:82-102  the 1 + rotations x speeds action table (host, once)

The per-step arithmetic that the reference does in Python for each of the 81 candidate
actions (propagate :104-129, rotate :217-252, compute_reward multi_human_rl.py:65-88) runs on
the GPU in sarl_*.hip; see policy/sarl.py.
"""
import itertools

import numpy as np
import torch.nn as nn


def mlp(input_dim, mlp_dims, last_relu=False):
    dims = [input_dim] + list(mlp_dims)
    layers = []
    last = len(dims) - 2
    for i, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
        layers.append(nn.Linear(a, b))
        if i != last or last_relu:
            layers.append(nn.ReLU())
    return nn.Sequential(*layers)


def build_action_space(v_pref, kinematics="holonomic", speed_samples=5, rotation_samples=16):
    """Returns (table [1 + R*S, 2] float64, speeds list, rotations ndarray).

    Row 0 is the zero action; then rotations-major, speeds-minor (itertools.product order).
    Holonomic rows are (vx, vy); unicycle rows are (v, r).
    """
    holonomic = kinematics == "holonomic"
    speeds = [(np.exp((i + 1) / speed_samples) - 1) / (np.e - 1) * v_pref for i in range(speed_samples)]
    if holonomic:
        rotations = np.linspace(0, 2 * np.pi, rotation_samples, endpoint=False)
    else:
        rotations = np.linspace(-np.pi / 4, np.pi / 4, rotation_samples)
    rows = [(0.0, 0.0)]
    for rot, spd in itertools.product(rotations, speeds):
        rows.append((spd * np.cos(rot), spd * np.sin(rot)) if holonomic else (spd, rot))
    return np.array(rows, dtype=np.float64), speeds, rotations


# ---------------------------------------------------------------------------------------------
import logging

import torch

from ..envs.policy.policy import Policy
from ..envs.utils.action import ActionRot, ActionXY
from ..envs.utils.state import FullState, ObservableState


def rotate(state, kinematics):
    """cadrl.py:217-252: [B,14] joint rows -> [B,13] agent-centric rows (torch, any device).

    Host-side use only (Explorer memory / `transform`); the look-ahead builds the same features inside
    sarl_value.hip.  Column order in : px py vx vy r gx gy v_pref theta | px1 py1 vx1 vy1 r1
    Column order out: dg v_pref theta r vx vy px1 py1 vx1 vy1 r1 da r+r1
    """
    px, py, vx, vy, r, gx, gy, vpref, theta, hx, hy, hvx, hvy, hr = state.unbind(1)
    dx, dy = gx - px, gy - py
    heading = torch.atan2(dy, dx)
    c, s = torch.cos(heading), torch.sin(heading)
    dg = torch.sqrt(dx * dx + dy * dy)
    th = theta - heading if kinematics == "unicycle" else torch.zeros_like(vpref)
    ox, oy = hx - px, hy - py
    da = torch.sqrt(ox * ox + oy * oy)
    return torch.stack([dg, vpref, th, r, vx * c + vy * s, vy * c - vx * s, ox * c + oy * s, oy * c - ox * s,
                        hvx * c + hvy * s, hvy * c - hvx * s, hr, da, r + hr], 1)


class CADRL(Policy):
    """Configuration / action-space / propagate surface shared by the value-based policies
    (cadrl.py:31-129).  CADRL's own single-human network is not on the path BASELINE.json names;
    SARL (policy/sarl.py) is, and inherits everything it needs from here."""

    def __init__(self):
        super().__init__()
        self.name = "CADRL"
        self.trainable = True
        self.multiagent_training = None
        self.kinematics = None
        self.epsilon = None
        self.gamma = None
        self.sampling = None
        self.speed_samples = None
        self.rotation_samples = None
        self.query_env = None
        self.action_space = None
        self.speeds = None
        self.rotations = None
        self.action_values = None
        self.with_om = None
        self.cell_num = None
        self.cell_size = None
        self.om_channel_size = None
        self.self_state_dim = 6
        self.human_state_dim = 7
        self.joint_state_dim = self.self_state_dim + self.human_state_dim
        self._action_table = None

    def set_common_parameters(self, config):
        self.gamma = config.getfloat("rl", "gamma")
        self.sampling = config.get("action_space", "sampling")
        self.speed_samples = config.getint("action_space", "speed_samples")
        self.rotation_samples = config.getint("action_space", "rotation_samples")
        self.query_env = config.getboolean("action_space", "query_env")
        self.cell_num = config.getint("om", "cell_num")
        self.cell_size = config.getfloat("om", "cell_size")
        self.om_channel_size = config.getint("om", "om_channel_size")

    def configure(self, config):
        raise NotImplementedError("CADRL's single-human value network is outside this build's scope; use 'sarl'")

    def set_device(self, device):
        self.device = device
        self.model.to(device)

    def set_epsilon(self, epsilon):
        self.epsilon = epsilon

    def build_action_space(self, v_pref):
        table, speeds, rotations = build_action_space(v_pref, self.kinematics, self.speed_samples, self.rotation_samples)
        make = ActionXY if self.kinematics == "holonomic" else ActionRot
        self.speeds, self.rotations = speeds, rotations
        self.action_space = [make(*row) for row in table.tolist()]
        self.action_space[0] = make(0, 0)
        self._action_table = table

    def propagate(self, state, action):
        """cadrl.py:104-129 (host value types; the batched look-ahead does this in sarl_value.hip)."""
        dt = self.time_step
        if isinstance(state, ObservableState):
            return ObservableState(state.px + action.vx * dt, state.py + action.vy * dt, action.vx, action.vy,
                                   state.radius)
        if not isinstance(state, FullState):
            raise ValueError("Type error")
        if self.kinematics == "holonomic":
            return FullState(state.px + action.vx * dt, state.py + action.vy * dt, action.vx, action.vy, state.radius,
                             state.gx, state.gy, state.v_pref, state.theta)
        import numpy as np
        th = state.theta + action.r
        nvx, nvy = action.v * np.cos(th), action.v * np.sin(th)
        return FullState(state.px + nvx * dt, state.py + nvy * dt, nvx, nvy, state.radius, state.gx, state.gy,
                         state.v_pref, th)

    def rotate(self, state):
        return rotate(state, self.kinematics)

    def transform(self, state):
        assert len(state.human_states) == 1
        row = torch.Tensor(state.self_state + state.human_states[0]).to(self.device)
        return self.rotate(row.unsqueeze(0)).squeeze(dim=0)
