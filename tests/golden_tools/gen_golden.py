#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the real reference in THIS container.

Run from the repo root:   python -m tests.golden_tools.gen_golden [--only g1,g2,...]

Nothing here travels to the GPU box except the arrays it writes.  Fixture families
(SURVEY.md section 8c):

  g1_reset      CrowdSim.reset scenarios           crowd_sim/envs/crowd_sim.py:94-217,261-323
  g2_step_*     CrowdSim.step / ModelCrowdSim.step crowd_sim.py:331-434, model_crowd_sim.py:347-441
  g3_p2s        point_to_segment_dist              crowd_sim/envs/utils/utils.py:4-26
  g4_actions    CADRL.build_action_space           crowd_nav/policy/cadrl.py:82-102
  g5_sarl       rotate / ValueNetwork / predict    cadrl.py:217-252, sarl.py:28-65, multi_human_rl.py:11-63
  g6_sgan       TrajectoryGenerator.forward        sgan/models.py:501-553, world_model.py:234-268
  g7_episode    hand-driven reset/act/step loops   crowd_nav/utils/explorer.py:54-125

ORCA-human fixtures (g2_step_orca, g7) run the reference's env code with tests/golden_tools/refshim's
rvo2 stand-in, i.e. THIS repo's C solver: they pin the plumbing around the solver, not the
solver (ORCA parity vs rvo2 stays unpinned).
"""
import argparse
import configparser
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.golden_tools import refshim  # noqa: E402

refshim.install()
OUT = os.path.join(ROOT, "tests", "golden")
REF = refshim.REFERENCE_ROOT

INFO_CODE = {"Nothing": 0, "Danger": 1, "ReachGoal": 2, "Collision": 3, "Timeout": 4}


def env_config(human_num=5, randomize=False, robot_visible=False, humans_policy="orca",
               train_val_sim="circle_crossing", test_sim="circle_crossing"):
    cfg = configparser.RawConfigParser()
    cfg.read(os.path.join(REF, "crowd_nav", "configs", "env.config"))
    cfg.set("env", "look_ahead_in_sim", "false")      # required by crowd_sim.py:81, missing from the shipped file
    cfg.set("env", "randomize_attributes", "true" if randomize else "false")
    cfg.set("sim", "human_num", str(human_num))
    cfg.set("sim", "train_val_sim", train_val_sim)
    cfg.set("sim", "test_sim", test_sim)
    cfg.set("robot", "visible", "true" if robot_visible else "false")
    cfg.set("humans", "policy", humans_policy)
    return cfg


def policy_config():
    cfg = configparser.RawConfigParser()
    cfg.read(os.path.join(REF, "crowd_nav", "configs", "policy.config"))
    return cfg


def make_env(cls_name="CrowdSim", robot_policy="orca", humans_policy="orca", **kw):
    import torch
    from crowd_sim.envs.crowd_sim import CrowdSim
    from crowd_sim.envs.model_crowd_sim import ModelCrowdSim
    from crowd_sim.envs.utils.robot import Robot
    from crowd_nav.policy.policy_factory import policy_factory
    cfg = env_config(humans_policy="orca", **kw)
    env = {"CrowdSim": CrowdSim, "ModelCrowdSim": ModelCrowdSim}[cls_name]()
    env.configure(cfg)                         # configure() only accepts 'orca' (crowd_sim.py:67-79)
    if humans_policy != "orca":
        cfg.set("humans", "policy", humans_policy)   # Human() reads it at reset (crowd_sim.py:166)
    robot = Robot(cfg, "robot")
    pol = policy_factory[robot_policy]()
    pol.configure(policy_config())
    if robot_policy == "sarl":
        pol.kinematics = "holonomic"            # drivers set it from --kinematics (train_model_based_sgan.py:115)
        pol.set_device(torch.device("cpu"))
        pol.set_phase("test")
    robot.set_policy(pol)
    env.set_robot(robot)
    pol.set_env(env)
    return env, robot, pol


def info_code(info):
    return INFO_CODE[type(info).__name__]


def full_state_rows(env):
    r = env.robot
    rob = [r.px, r.py, r.vx, r.vy, r.radius, r.gx, r.gy, r.v_pref, r.theta]
    hum = [[h.px, h.py, h.vx, h.vy, h.radius, h.gx, h.gy, h.v_pref, h.theta] for h in env.humans]
    return np.array(rob, np.float64), np.array(hum, np.float64).reshape(-1, 9)


# ---------------------------------------------------------------------------------------------
def g1_reset():
    recs = {}
    meta = []
    k = 0
    for robot_policy in ("sarl", "orca"):           # multiagent_training True / None
        for rule in ("circle_crossing", "square_crossing", "mixed"):
            for N in (5, 10):
                for rnd in (False, True):
                    env, robot, pol = make_env(robot_policy=robot_policy, human_num=N, randomize=rnd,
                                               train_val_sim=rule, test_sim=rule)
                    for phase in ("test", "val", "train"):
                        for case in range(12):
                            env.human_num = N       # 'mixed' overwrites it (crowd_sim.py:125)
                            env.reset(phase, case)
                            rob, hum = full_state_rows(env)
                            recs["rob_%d" % k] = rob
                            recs["hum_%d" % k] = hum
                            meta.append((robot_policy == "sarl", rule, N, rnd, phase, case, hum.shape[0]))
                            k += 1
    recs["meta_multiagent"] = np.array([m[0] for m in meta], np.uint8)
    recs["meta_rule"] = np.array([m[1] for m in meta])
    recs["meta_N"] = np.array([m[2] for m in meta], np.int32)
    recs["meta_randomize"] = np.array([m[3] for m in meta], np.uint8)
    recs["meta_phase"] = np.array([m[4] for m in meta])
    recs["meta_case"] = np.array([m[5] for m in meta], np.int32)
    recs["meta_nh"] = np.array([m[6] for m in meta], np.int32)
    # known answers quoted in SURVEY.md 8a (a9)
    env, _, _ = make_env(robot_policy="orca", human_num=5)
    env.reset("test", 0)
    assert env.humans[0].px == -2.6625559084662678 and env.humans[0].py == -2.837985290326649
    np.savez_compressed(os.path.join(OUT, "g1_reset.npz"), **recs)
    print("g1_reset: %d scenarios" % k)


def _random_scene(rng, env, N, mode):
    """Overwrite agent states with a random configuration that exercises every ladder branch."""
    from crowd_sim.envs.utils.action import ActionXY
    rob = env.robot
    rpx, rpy = rng.uniform(-4, 4, 2)
    if mode == "neargoal":
        gx, gy = rpx + rng.uniform(-0.5, 0.5), rpy + rng.uniform(-0.5, 0.5)
    else:
        gx, gy = rng.uniform(-4, 4, 2)
    rob.set(rpx, rpy, gx, gy, rng.uniform(-1, 1), rng.uniform(-1, 1), np.pi / 2)
    for h in env.humans:
        if mode == "close":
            ang, d = rng.uniform(0, 2 * np.pi), rng.uniform(0.45, 1.2)
            px, py = rpx + d * np.cos(ang), rpy + d * np.sin(ang)
        elif mode == "crowded":
            px, py = rng.uniform(-1.5, 1.5, 2)
        else:
            px, py = rng.uniform(-5, 5, 2)
        spd, va = rng.uniform(0, 1.2), rng.uniform(0, 2 * np.pi)
        h.set(px, py, rng.uniform(-5, 5), rng.uniform(-5, 5), spd * np.cos(va), spd * np.sin(va), 0.0)
        if env.randomize_attributes:
            h.sample_random_attributes()
    env.global_time = float(rng.choice([0.0, 3.25, 23.5, 23.75, 24.0, 24.25], p=[0.35, 0.35, 0.1, 0.1, 0.05, 0.05]))
    env.human_times = [0] * len(env.humans)
    sp, aa = rng.uniform(0, 1.0), rng.uniform(0, 2 * np.pi)
    return ActionXY(sp * np.cos(aa), sp * np.sin(aa))


def _g2(name, cls_name, humans_policy, robot_visible, n_samples, seed, unicycle=False):
    rng = np.random.RandomState(seed)
    rows = {k: [] for k in ("N", "update", "time", "rob_in", "hum_in", "act", "given_v", "reward", "done", "info",
                            "dmin", "rob_out", "hum_out", "obs", "time_out", "human_times")}
    for N in (5, 10, 3):
        for rnd in (False, True):
            env, robot, pol = make_env(cls_name, robot_policy="orca", humans_policy=humans_policy,
                                       human_num=N, randomize=rnd, robot_visible=robot_visible)
            if unicycle:
                robot.kinematics = "unicycle"          # Agent.kinematics, normally copied from the policy (agent.py:36)
            for s in range(n_samples):
                if cls_name == "CrowdSim":
                    env.reset("test", s % 7)        # fresh Human objects -> fresh ORCA sims
                else:
                    env.reset("test", no_random_gen=True)
                mode = ("spread", "close", "crowded", "neargoal")[s % 4]
                action = _random_scene(rng, env, N, mode)
                if unicycle:
                    from crowd_sim.envs.utils.action import ActionRot
                    env.robot.theta = rng.uniform(-7, 7)
                    action = ActionRot(rng.uniform(0, 1.0), rng.uniform(-np.pi / 4, np.pi / 4))
                for h in env.humans:
                    h.time_step = env.time_step
                    h.policy.time_step = env.time_step
                update = bool(s % 3 != 0)
                rob_in, hum_in = full_state_rows(env)
                t_in = env.global_time
                if cls_name == "ModelCrowdSim":
                    gv = rng.uniform(-1, 1, (N, 2))
                    ob, reward, done, info = env.step(action, update=update, new_v=gv.tolist())
                else:
                    gv = np.zeros((N, 2))
                    ob, reward, done, info = env.step(action, update=update)
                rob_out, hum_out = full_state_rows(env)
                rows["N"].append(N); rows["update"].append(update); rows["time"].append(t_in)
                rows["rob_in"].append(rob_in); rows["hum_in"].append(np.pad(hum_in, ((0, 10 - N), (0, 0))))
                rows["act"].append([action[0], action[1]]); rows["given_v"].append(np.pad(gv, ((0, 10 - N), (0, 0))))
                rows["reward"].append(reward); rows["done"].append(done); rows["info"].append(info_code(info))
                rows["dmin"].append(getattr(info, "min_dist", np.nan))
                rows["rob_out"].append(rob_out); rows["hum_out"].append(np.pad(hum_out, ((0, 10 - N), (0, 0))))
                obs = np.array([[o.px, o.py, o.vx, o.vy, o.radius] for o in ob], np.float64)
                rows["obs"].append(np.pad(obs, ((0, 10 - N), (0, 0))))
                rows["time_out"].append(env.global_time)
                ht = list(env.human_times) if cls_name == "CrowdSim" else [0] * N
                rows["human_times"].append(np.pad(np.array(ht, np.float64), (0, 10 - N)))
    out = {k: np.array(v) for k, v in rows.items()}
    out["robot_visible"] = np.array(robot_visible)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    codes, cnt = np.unique(out["info"], return_counts=True)
    print("%s: %d samples, info histogram %s" % (name, len(out["N"]), dict(zip(codes.tolist(), cnt.tolist()))))


def g2_step():
    _g2("g2_step_given", "ModelCrowdSim", "orca", False, 240, 11)
    _g2("g2_step_linear", "CrowdSim", "linear", False, 160, 12)
    _g2("g2_step_orca", "CrowdSim", "orca", False, 200, 13)
    _g2("g2_step_orca_visible", "CrowdSim", "orca", True, 120, 14)
    _g2("g2_step_unicycle", "ModelCrowdSim", "orca", False, 120, 15, unicycle=True)


def g3_p2s():
    from crowd_sim.envs.utils.utils import point_to_segment_dist
    rng = np.random.RandomState(3)
    args = rng.uniform(-3, 3, (4000, 6))
    args[:200, 2:4] = args[:200, 0:2]            # degenerate segment
    args[200:400, 4:6] = 0.0                     # query at the origin (the env's call shape)
    args[400] = (1, 1, 2, 3, 0, 0); args[401] = (-1, 1, 2, 1, 0, 0)
    out = np.array([float(point_to_segment_dist(*a)) for a in args])
    assert out[400] == 1.4142135623730951 and out[401] == 1.0
    # re-check the fma finding the C oracle relies on
    chk = np.array([float(np.linalg.norm((a[0], a[1]))) for a in args])
    np.savez_compressed(os.path.join(OUT, "g3_p2s.npz"), args=args, dist=out, norm01=chk)
    print("g3_p2s: %d" % len(out))


def g4_actions():
    from crowd_nav.policy.sarl import SARL
    res = {}
    for kin in ("holonomic", "unicycle"):
        for v_pref in (1.0, 0.7):
            p = SARL(); p.configure(policy_config()); p.kinematics = kin
            p.build_action_space(v_pref)
            res["%s_%g" % (kin, v_pref)] = np.array([[a[0], a[1]] for a in p.action_space], np.float64)
            res["speeds_%s_%g" % (kin, v_pref)] = np.array(p.speeds, np.float64)
            res["rotations_%s_%g" % (kin, v_pref)] = np.array(p.rotations, np.float64)
    a = res["holonomic_1"]
    assert res["speeds_holonomic_1"][0] == 0.12885124808584156
    assert tuple(a[6]) == (0.11904303084504313, 0.04930923788201555)
    np.savez_compressed(os.path.join(OUT, "g4_actions.npz"), **res)
    print("g4_actions ok, %d actions" % len(a))


FAMILIES = {"g1": g1_reset, "g2": g2_step, "g3": g3_p2s, "g4": g4_actions}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    try:
        from tests.golden_tools import gen_golden_nets
        FAMILIES.update(gen_golden_nets.FAMILIES)
    except ImportError:
        pass
    want = [w for w in args.only.split(",") if w] or list(FAMILIES)
    for w in want:
        FAMILIES[w]()


if __name__ == "__main__":
    main()
