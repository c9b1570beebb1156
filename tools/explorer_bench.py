#!/usr/bin/env python3
"""GPU box: per-step cost of the batched Explorer with a SARL robot (BASELINE config 3's shape), with and without the
training-time bookkeeping (update_memory: the rotated joint state of every step is stored, value targets at the end)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from modelcrowdnav_amd.rollout import VecExplorer  # noqa: E402
from modelcrowdnav_amd.utils.memory import ReplayMemory  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    E, N = 4096, 5
    env, _ = bench.build_env(E, N, 0, dev)
    env.detach_rollout()
    pol = bench._sarl_policy(dev, env.time_step)
    env.robot.set_policy(pol)
    mem = ReplayMemory(2000000, device=dev)
    ex = VecExplorer(env, env.robot, gamma=0.9, policy=pol, memory=mem)
    ex.update_target_model(pol.get_model())
    for upd in (False, True, False, True):
        pol.set_phase("train" if upd else "test")
        pol.set_epsilon(0.1 if upd else 0.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ex.run_k_episodes(E, "train" if upd else "test", update_memory=upd)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        # an untrained robot mostly times out: episodes last the full 100 steps (+ the 32-step polling granularity)
        print("update_memory=%s: %.1f ms for one episode per env, %d envs (~%.2f ms per step over ~128 steps incl. set-up / "
              "value targets), memory %d" % (upd, el * 1e3, E, el * 1e3 / 128, len(mem)))


def dropin():
    """What a reference driver gets after dropin.install(): crowd_nav.utils.explorer.Explorer on the E = 1 gym env,
    `run_k_episodes(500, 'test')` with a SARL robot (test.py:109, train.py:249) -- handed to the batched VecExplorer --
    against the same call kept on the sequential E = 1 loop (a few episodes, scaled)."""
    from modelcrowdnav_amd import configs
    from modelcrowdnav_amd.envs import CrowdSim
    from modelcrowdnav_amd.envs.utils.robot import Robot
    from modelcrowdnav_amd.utils.explorer import Explorer
    dev = torch.device("cuda", 0)
    cfg = configs.env_config(**{"sim.human_num": 5})
    env = CrowdSim()
    env.configure(cfg)
    robot = Robot(cfg, "robot")
    pol = bench._sarl_policy(dev, 0.25)
    robot.set_policy(pol)
    env.set_robot(robot)
    pol.set_env(env)
    ex = Explorer(env, robot, dev, gamma=0.9)
    k = 500
    ex.run_k_episodes(k, "test")                                   # builds the batched twin, loads code objects
    env.case_counter["test"] = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out_b = ex.run_k_episodes(k, "test")
    torch.cuda.synchronize()
    tb = time.perf_counter() - t0
    assert ex.last_run_batched
    ex.batched = False
    ks = 4
    env.case_counter["test"] = 0
    ex.run_k_episodes(1, "test")
    env.case_counter["test"] = 0
    t0 = time.perf_counter()
    out_s = ex.run_k_episodes(ks, "test")
    ts = time.perf_counter() - t0
    print("drop-in Explorer.run_k_episodes(%d, 'test'), SARL robot, E = 1 gym env: batched %.3f s (%.2f ms per episode); "
          "sequential %.3f s for %d episodes (%.1f ms per episode) -> %.0f x" %
          (k, tb, tb / k * 1e3, ts, ks, ts / ks * 1e3, (ts / ks) / (tb / k)))
    print("  returned (batched):    ", out_b)
    print("  returned (sequential, first %d cases):" % ks, out_s)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "dropin":
        dropin()
    else:
        main()
