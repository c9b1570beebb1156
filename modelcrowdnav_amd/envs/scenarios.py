"""Seeded scenario generation for CrowdSim.reset (host side, bit-exact with the reference).

Reference: crowd_sim/envs/crowd_sim.py:94-217 (generators), :261-323 (reset),
crowd_sim/envs/model_crowd_sim.py:160-227 (variants that also draw an initial velocity).

Why this stays on the host: the reference draws from numpy's global MT19937 stream with
sequential rejection sampling (3 draws per circle attempt), so a scenario is a function of
(seed, draw order).  Reproducing it bit for bit means consuming the same stream in the same
order; it runs once per episode, not per step, and its output is uploaded to HBM (or kept as
a resident pool that env_step.hip resets from).  `rng` is any object with numpy's
`random()` / `uniform()` methods: the `numpy.random` module itself reproduces the reference's
global-stream behaviour, a `numpy.random.RandomState(seed)` gives an independent per-env stream
with identical values.

A scenario is a float64 array [n_humans, 9]: px, py, gx, gy, vx, vy, theta, radius, v_pref.
"""
import numpy as np
from numpy.linalg import norm

PX, PY, GX, GY, VX, VY, TH, RAD, VPREF = range(9)


class ScenarioSpec(object):
    """The scalar knobs the generators read (env.config [sim]/[humans]/[robot]/[reward])."""

    def __init__(self, circle_radius=4.0, square_width=10.0, discomfort_dist=0.2, human_radius=0.3,
                 human_v_pref=1.0, robot_radius=0.3, randomize_attributes=False, init_velocity=False):
        self.circle_radius = circle_radius
        self.square_width = square_width
        self.discomfort_dist = discomfort_dist
        self.human_radius = human_radius
        self.human_v_pref = human_v_pref
        self.robot_radius = robot_radius
        self.randomize_attributes = randomize_attributes
        self.init_velocity = init_velocity     # ModelCrowdSim flavour (model_crowd_sim.py:183-192)

    def robot_row(self):
        r = self.circle_radius
        return np.array([0.0, -r, 0.0, r, 0.0, 0.0, np.pi / 2, self.robot_radius, 1.0])


def _attrs(spec, rng):
    v_pref, radius = spec.human_v_pref, spec.human_radius
    if spec.randomize_attributes:               # agent.py:39-45: v_pref first, then radius
        v_pref = rng.uniform(0.5, 1.5)
        radius = rng.uniform(0.3, 0.5)
    return radius, v_pref


def _init_v(px, py, gx, gy, v_pref):
    vx, vy = gx - px, gy - py
    vmax = abs(vx)
    if vmax < abs(vy):
        vmax = abs(vy)
    return v_pref * vx / vmax, v_pref * vy / vmax


def _clear_of(px, py, placed, radius, spec, goals_too):
    for q in placed:
        gap = radius + q[RAD] + spec.discomfort_dist
        if norm((px - q[PX], py - q[PY])) < gap:
            return False
        if goals_too and norm((px - q[GX], py - q[GY])) < gap:
            return False
    return True


def circle_crossing_human(spec, rng, placed):
    """crowd_sim.py:165-186: start on a noisy circle, goal at the antipode; keep clear of every
    earlier agent's position AND goal."""
    radius, v_pref = _attrs(spec, rng)
    while True:
        angle = rng.random() * np.pi * 2
        nx = (rng.random() - 0.5) * v_pref
        ny = (rng.random() - 0.5) * v_pref
        px = spec.circle_radius * np.cos(angle) + nx
        py = spec.circle_radius * np.sin(angle) + ny
        if _clear_of(px, py, placed, radius, spec, goals_too=True):
            break
    vx = vy = 0.0
    if spec.init_velocity:
        vx, vy = _init_v(px, py, -px, -py, v_pref)
    return np.array([px, py, -px, -py, vx, vy, 0.0, radius, v_pref])


def square_crossing_human(spec, rng, placed):
    """crowd_sim.py:188-217: start on one side of the y axis, goal on the other."""
    radius, v_pref = _attrs(spec, rng)
    sign = -1 if rng.random() > 0.5 else 1
    w = spec.square_width
    while True:
        px = rng.random() * w * 0.5 * sign
        py = (rng.random() - 0.5) * w
        if _clear_of(px, py, placed, radius, spec, goals_too=False):
            break
    while True:
        gx = rng.random() * w * 0.5 * -sign
        gy = (rng.random() - 0.5) * w
        ok = True
        for q in placed:
            if norm((gx - q[GX], gy - q[GY])) < radius + q[RAD] + spec.discomfort_dist:
                ok = False
                break
        if ok:
            break
    vx = vy = 0.0
    if spec.init_velocity:
        vx, vy = _init_v(px, py, -px, -py, v_pref)     # sic: model_crowd_sim.py:225 aims at (-px,-py)
    return np.array([px, py, gx, gy, vx, vy, 0.0, radius, v_pref])


_STATIC_MIX = ((0, 0.05), (1, 0.2), (2, 0.2), (3, 0.3), (4, 0.1), (5, 0.15))
_DYNAMIC_MIX = ((1, 0.3), (2, 0.3), (3, 0.2), (4, 0.1), (5, 0.1))


def generate(spec, rng, human_num, rule):
    """crowd_sim.py:94-163.  Returns [n,9]; n may differ from human_num under rule 'mixed'."""
    robot = spec.robot_row()
    placed = [robot]
    if rule == "square_crossing":
        for _ in range(human_num):
            placed.append(square_crossing_human(spec, rng, placed))
    elif rule == "circle_crossing":
        for _ in range(human_num):
            placed.append(circle_crossing_human(spec, rng, placed))
    elif rule == "mixed":
        static = rng.random() < 0.2
        prob = rng.random()
        for n, share in (_STATIC_MIX if static else _DYNAMIC_MIX):
            if prob - share <= 0:
                human_num = n
                break
            prob -= share
        if static:
            width, height = 4, 8
            if human_num == 0:
                placed.append(np.array([0.0, -10.0, 0.0, -10.0, 0.0, 0.0, 0.0, spec.human_radius, spec.human_v_pref]))
            for _ in range(human_num):
                sign = -1 if rng.random() > 0.5 else 1
                while True:
                    px = rng.random() * width * 0.5 * sign
                    py = (rng.random() - 0.5) * height
                    if _clear_of(px, py, placed, spec.human_radius, spec, goals_too=False):
                        break
                placed.append(np.array([px, py, px, py, 0.0, 0.0, 0.0, spec.human_radius, spec.human_v_pref]))
        else:
            for i in range(human_num):
                maker = circle_crossing_human if i < 2 else square_crossing_human
                placed.append(maker(spec, rng, placed))
    else:
        raise ValueError("Rule doesn't exist")
    return np.array(placed[1:], dtype=np.float64).reshape(-1, 9)


SEED_OFFSET = {"train": 2000, "val": 0, "test": 1000}   # crowd_sim.py:68,282-283 (case_capacity val+test, 0, val)


def scenario_for_case(spec, phase, case, human_num, rule):
    """One seeded scenario on an independent stream: identical values to
    `np.random.seed(offset + case)` followed by the reference generator (crowd_sim.py:286-292)."""
    rs = np.random.RandomState(SEED_OFFSET[phase] + int(case))
    return generate(spec, rs, human_num, rule)


def scenario_pool(spec, phase, cases, human_num, rule):
    """Stack scenarios for a list of cases -> [P, N, 9] (fixed-N rules only)."""
    rows = [scenario_for_case(spec, phase, c, human_num, rule) for c in cases]
    for r in rows:
        if r.shape[0] != human_num:
            raise ValueError("rule %r produced a varying human count; batched envs need a fixed N" % rule)
    return np.stack(rows, 0)
