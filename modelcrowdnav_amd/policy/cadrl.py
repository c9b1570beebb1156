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
