"""What a SARL-driven env step (BASELINE config 3) consists of: the look-ahead launch, the env step, and the rest.
    python tools/sarl_step_probe.py [--humans 5]      (run under rocprofv3 --kernel-trace --stats for the kernel list)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--humans", type=int, default=5)
    ap.add_argument("--iters", type=int, default=50)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    E, N = 4096, a.humans
    env, _ = bench.build_env(E, N, 0, dev)
    pol = bench._sarl_policy(dev, env.time_step)
    act, _ = pol.predict_batch(env)
    act = act.clone()

    def both():
        x, _ = pol.predict_batch(env)
        env.step(x)
    t_pred = bench._timed(lambda: pol.predict_batch(env), a.iters)
    t_step = bench._timed(lambda: env.step(act), a.iters)
    t_both = bench._timed(both, a.iters)
    g = torch.cuda.CUDAGraph()
    both(); torch.cuda.synchronize()
    try:
        with torch.cuda.graph(g):
            for _ in range(10):
                both()
        t_graph = bench._timed(g.replay, 5) / 10
    except Exception as ex:  # noqa: BLE001
        t_graph = float("nan")
        print("graph capture failed:", ex)
    print("N=%d: predict_batch %.4f ms, env.step %.4f ms, both %.4f ms (sum %.4f), both inside one hipGraph %.4f ms"
          % (N, t_pred, t_step, t_both, t_pred + t_step, t_graph))


if __name__ == "__main__":
    main()
