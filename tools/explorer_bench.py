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


if __name__ == "__main__":
    main()
