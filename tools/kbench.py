#!/usr/bin/env python3
"""Kernel micro-benchmark (GPU box): per-launch time of mcn_env_step at several batch sizes and
human-policy modes, measured with HIP events around a hipGraph of back-to-back launches.
    python tools/kbench.py [--humans 5] [--sizes 4096,65536,1048576] [--modes orca,given]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def sgan_bench(E, N, iters):
    """One SGANWorld call (ring push + encoder + pool-net + decoder) for E scenes of N pedestrians."""
    from modelcrowdnav_amd.policy.world_model import VecSGANWorld, generator_from_arrays
    dev = torch.device("cuda", 0)
    gen = generator_from_arrays(np.load(os.path.join(ROOT, "tests", "golden", "g6_sgan.npz")), "p", dev)
    world = VecSGANWorld(gen, E, N, dev, time_step=0.25, seed=0)
    g = torch.Generator(device="cpu").manual_seed(0)
    pos = (torch.rand(E, N, 2, dtype=torch.float64, generator=g) * 8 - 4).to(dev)
    vel = (torch.rand(E, N, 2, dtype=torch.float64, generator=g) - 0.5).to(dev)
    world.init_constant_velocity(pos, vel)
    for _ in range(3):
        world(pos)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        world(pos)
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / iters
    tf = 0.69e6 * E * N / ms / 1e9
    print("SGAN step N=%d E=%d: %.1f us/call  %.1f TFLOP/s algorithmic (%.1f %% of the fp32 MFMA peak %.1f)" % (
        N, E, ms * 1e3, tf, 100 * tf / bench.MFMA_F32_PEAK_TFLOPS, bench.MFMA_F32_PEAK_TFLOPS), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--humans", type=str, default="5", help="humans per env (a comma list with --sarl)")
    ap.add_argument("--sizes", default="4096,65536,1048576")
    ap.add_argument("--modes", default="orca,given")
    ap.add_argument("--visible", action="store_true")
    ap.add_argument("--no-hh", action="store_true", help="no human-human overlap count (ModelCrowdSim.step does not count)")
    ap.add_argument("--pair-stream", type=int, default=-1, help="mcn_tuning.pair_stream")
    ap.add_argument("--lp3-defer", type=int, default=-1, help="mcn_tuning.lp3_defer")
    ap.add_argument("--step-block", type=int, default=-1, help="mcn_tuning.step_block (64 / 256)")
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--sarl", action="store_true")
    ap.add_argument("--rollout", type=int, default=0, help="time mcn_env_rollout with this many steps per launch")
    ap.add_argument("--sgan", action="store_true", help="time mcn_sgan_step (shipped pool-net weights) at --sizes x --humans")
    a = ap.parse_args()
    if a.sgan:
        for N in [int(x) for x in str(a.humans).split(",")]:
            for E in [int(x) for x in a.sizes.split(",")]:
                sgan_bench(E, N, a.iters)
        return
    if not a.sarl:
        a.humans = int(a.humans)
    if a.rollout:
        rollout_bench(a)
        return
    if a.sarl:
        for N in [int(x) for x in str(a.humans).split(",")]:
            sarl_bench(4096, N)
        return
    dev = torch.device("cuda", 0)
    N = a.humans
    if a.step_block > 0:
        from modelcrowdnav_amd import _hip
        _hip.set_tuning(step_block=a.step_block)
    if a.lp3_defer >= 0:                       # before the env allocates (or skips) its 3-D-LP queue
        from modelcrowdnav_amd import _hip
        _hip.set_tuning(lp3_defer=a.lp3_defer)
    for E in [int(x) for x in a.sizes.split(",")]:
        env, _ = bench.build_env(E, N, 0, dev)
        env.robot.visible = a.visible
        if a.no_hh:
            env.count_hh = False
        if a.pair_stream >= 0:
            from modelcrowdnav_amd import _hip
            _hip.set_tuning(pair_stream=a.pair_stream)
        if a.lp3_defer >= 0:
            from modelcrowdnav_amd import _hip
            _hip.set_tuning(lp3_defer=a.lp3_defer)
        acts = bench.make_actions(16, E, E, 0, dev)
        gv = torch.rand(E, N, 2, dtype=torch.float64, device=dev) - 0.5
        for mode in a.modes.split(","):
            g = gv if mode.startswith("given") else None
            if mode.endswith("-noroll"):
                env.detach_rollout()        # no Explorer bookkeeping / auto-reset: fewer per-env streams
            if mode.endswith("-nopool"):
                env.detach_rollout()
                env.attach_rollout(gamma=0.9)   # Explorer record only, no restart from the pool
            for t in range(8):
                env.step(acts[t], given_v=g)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for t in range(a.iters):
                    env.step(acts[t % 16], given_v=g)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            best = 1e9
            for rep in range(3):
                s.record(); graph.replay(); e.record(); torch.cuda.synchronize()
                best = min(best, s.elapsed_time(e) / a.iters)
            nb = bench.pairwise_bytes_per_env_step(N) if mode.startswith("given") else bench.algorithmic_bytes_per_env_step(N)
            print("N=%d E=%8d mode=%-12s  %9.2f us/launch  %8.1f M env-steps/s  %7.1f GB/s (%.1f%% of 8 TB/s)" % (
                N, E, mode, best * 1e3, E / best / 1e3, nb * E / best / 1e6, nb * E / best / 1e6 / 80.0))
        del env




def rollout_bench(a):
    """Per-step time of the fused T-step launch (mcn_set_tuning(rollout_fused=1) forces it at any batch size)."""
    dev = torch.device("cuda", 0)
    N, T = a.humans, a.rollout
    from modelcrowdnav_amd import _hip
    _hip.set_tuning(rollout_fused=1)
    for E in [int(x) for x in a.sizes.split(",")]:
        env, _ = bench.build_env(E, N, 0, dev)
        env.robot.visible = a.visible
        acts = bench.make_actions(T, E, E, 0, dev)
        env.rollout(acts)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for rep in range(5):
            s.record(); env.rollout(acts); e.record(); torch.cuda.synchronize()
            best = min(best, s.elapsed_time(e) / T)
        print("N=%d E=%8d rollout T=%d  %9.3f us/step  %8.1f M env-steps/s" % (N, E, T, best * 1e3, E / best / 1e3))
        del env


def sarl_bench(E=4096, N=5, iters=5):
    """mcn_sarl_lookahead alone and SARL-driven env steps (BASELINE config 3)."""
    from modelcrowdnav_amd import configs
    from modelcrowdnav_amd.policy.sarl import SARL
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    pol = SARL(); pol.configure(configs.policy_config()); pol.kinematics = "holonomic"
    pol.set_device(dev); pol.set_phase("test"); pol.time_step = 0.25
    env, _ = bench.build_env(E, N, 0, dev)
    for _ in range(2):
        a, b = pol.predict_batch(env)
        env.step(a)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        pol.predict_batch(env)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    r = bench._sarl_roofline(E, N, ms)
    print("SARL lookahead N=%d E=%d: %.3f ms/launch  %.1f TFLOP/s executed = %.3f of the %.1f peak [%s]; float32-MFMA "
          "equivalent %.1f TFLOP/s = %.3f of the fp32 MFMA peak 157.3 (%.1f by the reference's FLOP count)  %.3f M env-steps/s" % (
              N, E, ms, r["achieved"], r["frac"], r["peak"], "bf16x3" if bench.sarl_uses_x3() else "f32", r["f32_mfma_equivalent_rate"],
              r["f32_mfma_equivalent_rate_over_f32_peak"], r["reference_flop_rate"], E / ms / 1e3))
    s.record()
    for _ in range(iters):
        a, b = pol.predict_batch(env)
        env.step(a)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print("SARL-driven env step N=%d E=%d: %.3f ms/step  %.3f M env-steps/s" % (N, E, ms, E / ms / 1e3))


if __name__ == "__main__":
    main()
