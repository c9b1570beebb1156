"""CPU baseline of SURVEY 8(d): a scalar, object-per-agent Python env step -- the SHAPE of the reference's
CrowdSim.step (crowd_sim/envs/crowd_sim.py:331-434: one Python object per agent, one observation list per human, one
ORCA solve per human through a native call, Python loops for the swept-circle test, the overlap count, the reward
ladder and the integration), with oracle/mcn_oracle.c's ORCA solver standing in for the absent rvo2 module
(orca.py:95-129's parameters).  Timed only by bench.py's cpu_baseline leg (kind "python-scalar") and checked against
the C oracle's batched step by tests/test_bench_helpers.py; never part of the product path.

The value types are the drop-in surface's own files (envs/utils/state.py, action.py, utils.py), loaded by path so that
this process imports neither torch nor the HIP library (bench.py runs it in child processes that must not touch the GPU).
"""
import importlib.util
import math
import os

import numpy as np

from oracle import cport

_UTILS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "modelcrowdnav_amd", "envs", "utils")


def _load(name):
    spec = importlib.util.spec_from_file_location("mcn_scalar_" + name, os.path.join(_UTILS, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_state, _action, _geom = _load("state"), _load("action"), _load("utils")
ObservableState, FullState, ActionXY = _state.ObservableState, _state.FullState, _action.ActionXY
NOTHING, DANGER, REACHGOAL, COLLISION, TIMEOUT = range(5)        # utils/info.py codes as the oracle numbers them


class ScalarAgent(object):
    def __init__(self, px, py, gx, gy, radius, v_pref):
        self.px, self.py, self.gx, self.gy, self.vx, self.vy = px, py, gx, gy, 0.0, 0.0
        self.radius, self.v_pref = radius, v_pref

    def observable(self):
        return ObservableState(self.px, self.py, self.vx, self.vy, self.radius)

    def orca(self, others, cfg):
        """orca.py:95-129: self with radius + 0.01 + safety_space and max speed v_pref, the others with max speed 1,
        preferred velocity = the raw goal vector; one native solve, float32 inside."""
        pad = 0.01 + cfg.orca_safety_space
        vx, vy = cport.orca_agent((self.px, self.py), (self.vx, self.vy), self.radius + pad, self.v_pref,
                                  (self.gx - self.px, self.gy - self.py),
                                  [o.position for o in others], [o.velocity for o in others],
                                  [o.radius + pad for o in others], cfg.orca_neighbor_dist, cfg.orca_max_neighbors,
                                  cfg.orca_time_horizon, cfg.time_step)
        return ActionXY(float(vx), float(vy))

    def move(self, action, dt):
        self.px, self.py, self.vx, self.vy = self.px + action.vx * dt, self.py + action.vy * dt, action.vx, action.vy


class ScalarCrowdSim(object):
    """One environment: holonomic invisible robot, ORCA humans (BASELINE configs[1])."""

    def __init__(self, scenario, cfg):
        self.cfg, self.scenario = cfg, np.asarray(scenario, dtype=np.float64)
        self.reset()

    def reset(self):
        self.humans = [ScalarAgent(r[0], r[1], r[2], r[3], r[7], r[8]) for r in self.scenario]
        self.robot = ScalarAgent(0.0, -4.0, 0.0, 4.0, 0.3, 1.0)
        self.global_time, self.human_times = 0.0, [0.0] * len(self.humans)

    def step(self, action):
        c, dt, robot = self.cfg, self.cfg.time_step, self.robot
        human_actions = []
        for human in self.humans:
            ob = [other.observable() for other in self.humans if other is not human]
            human_actions.append(human.orca(ob, c))
        dmin, collision = float("inf"), False
        for human in self.humans:
            px, py = human.px - robot.px, human.py - robot.py
            vx, vy = human.vx - action.vx, human.vy - action.vy
            gap = _geom.point_to_segment_dist(px, py, px + vx * dt, py + vy * dt, 0, 0) - human.radius - robot.radius
            if gap < 0:
                collision = True
                break
            dmin = min(dmin, gap)
        overlaps = 0
        for i, a in enumerate(self.humans):
            for b in self.humans[i + 1:]:
                overlaps += ((a.px - b.px) ** 2 + (a.py - b.py) ** 2) ** 0.5 - a.radius - b.radius < 0
        ex, ey = robot.px + action.vx * dt, robot.py + action.vy * dt
        reaching = math.hypot(ex - robot.gx, ey - robot.gy) < robot.radius
        if self.global_time >= c.time_limit - 1:
            reward, done, info = 0.0, True, TIMEOUT
        elif collision:
            reward, done, info = c.collision_penalty, True, COLLISION
        elif reaching:
            reward, done, info = c.success_reward, True, REACHGOAL
        elif dmin < c.discomfort_dist:
            reward, done, info = (dmin - c.discomfort_dist) * c.discomfort_penalty_factor * dt, False, DANGER
        else:
            reward, done, info = 0.0, False, NOTHING
        robot.move(action, dt)
        for human, act in zip(self.humans, human_actions):
            human.move(act, dt)
        self.global_time += dt
        for i, human in enumerate(self.humans):
            if self.human_times[i] == 0 and math.hypot(human.px - human.gx, human.py - human.gy) < human.radius:
                self.human_times[i] = self.global_time
        return [h.observable() for h in self.humans], reward, done, info, overlaps


def run(scenarios, table, seconds, seed=0):
    """Steps ONE env at a time (the reference is sequential), auto-reset on done, cycling over `scenarios`.
    Returns (env_steps, elapsed_s)."""
    import time
    cfg, rng = cport.default_cfg(), np.random.RandomState(seed)
    envs = [ScalarCrowdSim(s, cfg) for s in scenarios]
    steps, t0 = 0, time.perf_counter()
    while True:
        for env in envs:
            a = table[rng.randint(0, len(table))]
            if env.step(ActionXY(float(a[0]), float(a[1])))[2]:
                env.reset()
            steps += 1
        el = time.perf_counter() - t0
        if el >= seconds:
            return steps, el


if __name__ == "__main__":
    import sys
    z = np.load(sys.argv[1])
    n, el = run(z["sc"][:int(sys.argv[3]) if len(sys.argv) > 3 else 16], z["tab"], float(sys.argv[2]))
    print("%d %.6f" % (n, el))
