#!/usr/bin/env python3
"""GPU box: steps per second of the E = 1 gym view (BASELINE configs[0] plumbing: CrowdSim + ORCA robot driven like
test.py --policy orca), i.e. what a caller of the reference's one-env surface sees.  python tools/e1_time.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from modelcrowdnav_amd import configs  # noqa: E402
from modelcrowdnav_amd.envs import CrowdSim  # noqa: E402
from modelcrowdnav_amd.envs.utils.robot import Robot  # noqa: E402
from modelcrowdnav_amd.envs.policy.policy_factory import policy_factory  # noqa: E402

cfg = configs.env_config()
env = CrowdSim()
env.configure(cfg)
robot = Robot(cfg, "robot")
pol = policy_factory["orca"]()
pol.configure(cfg)
robot.set_policy(pol)
env.set_robot(robot)
pol.set_phase("test"); pol.set_device(torch.device("cuda", 0)); pol.set_env(env)
steps, t_act, t_step = 0, 0.0, 0.0
for case in range(12):
    ob = env.reset("test", case)
    done = False
    while not done:
        t0 = time.perf_counter()
        a = robot.act(ob)
        t1 = time.perf_counter()
        ob, r, done, info = env.step(a)
        t2 = time.perf_counter()
        if case >= 2:                          # the first episodes warm everything up
            steps += 1; t_act += t1 - t0; t_step += t2 - t1
print("E = 1 CrowdSim, ORCA robot: %d steps, robot.act %.1f us, env.step %.1f us -> %.0f env-steps/s"
      % (steps, t_act / steps * 1e6, t_step / steps * 1e6, steps / (t_act + t_step)))
