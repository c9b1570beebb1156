#!/usr/bin/env python3
"""GPU box: per-step time of mcn_env_rollout for 6-10 ORCA humans -- one looped launch (env_step_loop_kernel) against
the T single-step launches `mcn_tuning.rollout_fused = 0` keeps.  python tools/loop_time.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from tests import helpers as H
from modelcrowdnav_amd import _hip
def run(N, E=4096, T=400):
    rng = np.random.RandomState(0)
    sp, aa = rng.uniform(0, 1, (T, E)), rng.uniform(0, 2*np.pi, (T, E))
    res = {}
    for fused in (1, 0):
        env = H.make_vec_env(E, N)
        env.track_human_times = False; env.export_human_actions = False
        env.reset("test", test_cases=[i % 500 for i in range(E)])
        from modelcrowdnav_amd.envs import scenarios as S
        pool = S.scenario_pool(env.spec(), "test", list(range(500)), N, "circle_crossing")
        env.attach_rollout(0.9, pool=pool, case_stride=1, first_cases=np.arange(E) % 500, fin_slots=2)
        acts = torch.from_numpy(np.stack([sp*np.cos(aa), sp*np.sin(aa)], -1)).to(env.device)
        _hip.set_tuning(rollout_fused=fused)
        env.rollout(acts[:50]); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for rep in range(3):
            env.rollout(acts)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (3 * T)
        res[fused] = dt * 1e6
        _hip.set_tuning(rollout_fused=-1)
    print("N=%d E=%d: one launch %.2f us/step, T launches %.2f us/step" % (N, E, res[1], res[0]))
for N in (10, 7, 6):
    run(N)
run(10, E=16384, T=100)
