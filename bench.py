#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the batched CrowdSim rollout hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1]): 4096 envs x 5 humans per GPU, ORCA humans, robot actions
drawn uniformly from the 81-entry action table by torch.Generator(seed=0), robot invisible,
scenarios = test cases 1000 + (global_env_id mod 500), auto-reset on done from an HBM-resident
pool of the same 500 scenarios, Explorer-style discounted returns accumulated in-kernel.
A "step" is one CrowdSim.step of the whole batch.  The robot's actions are a pre-drawn sequence, so the K
steps go to the device as ceil(K / 1000) mcn_env_rollout launches (env state in registers between the steps
of a launch; --steps-per-launch 1 = one mcn_env_step launch per step, also reported as single_step_launch).
Inputs (states, the [K, E, 2] action tensor) are resident in HBM before the timed region; the launches are
replayed from one hipGraph.  When K steps take less than ~20 ms (the driver's --steps 20 is ~0.08 ms) the timed
region holds R back-to-back passes of the same K steps (R chosen from an untimed probe pass, reported as
config.replays; all R * K steps are executed on every env, none skipped) and ms_per_step = elapsed / (K * R).
(Measured and dropped: stepping the shard as P independent sub-batches on P streams of one graph does not hide the
straggling wavefronts of a launch -- a sub-batch launch is as latency-bound as the whole one.)
Envs shard across ranks with no per-step communication (weak scaling); at the end of the
rollout one RCCL all_gather collects episode returns + outcome codes.

Prints ONE JSON line (rank 0).  Extra objects: roofline (dominant kernel, live HIP-event timing),
roofline_sweep (same kernel at larger batches, where HBM rather than launch latency bounds it),
cpu_baseline (the C oracle on this host's cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s


def algorithmic_bytes_per_env_step(N):
    """(17 + 12 N) float64 scalars + 6 B of masks/counters."""
    return (17 + 12 * N) * 8 + 6


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    ap.add_argument("--humans", type=int, default=5)
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--steps-per-launch", type=int, default=1000,
                    help="env steps handed to one mcn_env_rollout call (the action sequence is known up front); "
                         "1 = one mcn_env_step launch per step")
    ap.add_argument("--sweep", type=str, default="65536,1048576,4194304", help="extra batch sizes for roofline_sweep")
    ap.add_argument("--no-sweep", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the SARL / SGAN configurations")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    ap.add_argument("--replays", type=int, default=0,
                    help="passes of the K steps inside the timed region (0 = as many as fill --min-timed-ms)")
    ap.add_argument("--no-merge", action="store_true",
                    help="keep one launch per pass of the K steps instead of merging consecutive passes of the timed "
                         "region into launches of up to --steps-per-launch steps")
    ap.add_argument("--force-collective", action="store_true",
                    help="run the N > 1 branches (process group, the one all_gather of episode records, the MAX / SUM "
                         "all-reduces) with a single rank too: exercises RCCL on a one-GPU box")
    ap.add_argument("--min-timed-ms", type=float, default=20.0,
                    help="the one-off costs of a timed region (sync, the final gather with several ranks) stay below ~2 %% of it")
    return ap.parse_args()


def build_env(E, N, rank_offset, device):
    from modelcrowdnav_amd import configs
    from modelcrowdnav_amd.envs.crowd_sim import VecCrowdSim
    from modelcrowdnav_amd.envs.utils.robot import Robot
    from modelcrowdnav_amd.envs.policy.policy_factory import policy_factory
    from modelcrowdnav_amd.envs import scenarios as S
    cfg = configs.env_config(**{"sim.human_num": N})
    env = VecCrowdSim(E, device)
    env.configure(cfg)
    robot = Robot(cfg, "robot")
    pol = policy_factory["orca"]()          # carrier only: actions come from the random table
    pol.multiagent_training = True
    robot.set_policy(pol)
    env.set_robot(robot)
    pool = S.scenario_pool(env.spec(), "test", range(500), N, "circle_crossing")      # [500,N,9]
    ids = (rank_offset + np.arange(E)) % 500
    env.load_scenarios(pool[ids])
    env.attach_rollout(gamma=0.9, pool=pool, case_stride=1, first_cases=(ids + 1) % 500)
    return env, pool


def action_table(v_pref=1.0):
    """The 81 holonomic actions (cadrl.py:82-102), built by the product's own policy code."""
    from modelcrowdnav_amd.policy.cadrl import build_action_space
    return torch.from_numpy(build_action_space(v_pref, "holonomic", 5, 16)[0])


def make_actions(steps, E, E_total, col0, device):
    """[steps, E, 2] float64: uniform draws over the 81-entry table, generator seed 0; drawn for the
    whole job and sliced per rank so the result does not depend on the partition."""
    gen = torch.Generator(device="cpu")
    gen.manual_seed(0)
    idx = torch.randint(0, 81, (steps, E_total), generator=gen, dtype=torch.int64)[:, col0:col0 + E]
    tab = action_table()
    return tab[idx].to(device).contiguous()


def time_kernel_events(env, acts, n, given_v=None, reps=3):
    """Average device duration of one mcn_env_step launch: HIP events (torch.cuda.Event on the launch
    stream, i.e. the stream handed to the C ABI) around a hipGraph of n back-to-back launches of that
    kernel and nothing else.  Returns (mean, best) ms per launch over `reps` replays."""
    for t in range(4):
        env.step(acts[t % acts.shape[0]], given_v=given_v)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for t in range(n):
            env.step(acts[t % acts.shape[0]], given_v=given_v)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    per = []
    for _ in range(reps):
        s.record()
        graph.replay()
        e.record()
        torch.cuda.synchronize()
        per.append(s.elapsed_time(e) / n)
    return float(np.mean(per)), float(np.min(per))


def pairwise_bytes_per_env_step(N):
    """Given-velocity (ModelCrowdSim.step / pairwise-only) variant: it needs no goals and no v_pref.
    read  robot px,py,gx,gy,r + action 2 + t 1 = 8, humans px,py,vx,vy,r + given vx,vy = 7N;
    write robot px,py,vx,vy + t = 5, humans px,py,vx,vy = 4N, reward + dmin = 2; + 6 B masks/counters."""
    return (15 + 11 * N) * 8 + 6


_PMC = None
_MODES = {0: "orca", 1: "linear", 2: "given"}          # include/mcn.h: MCN_HUMANS_*


def classify_kernel(name, run_humans=None):
    """(family, humans, mode, lanes per human) of an mcn:: kernel from its name and template arguments -- never from
    a substring guess: env_step_kernel<BLOCK, NT, VIS, MODE, HH>; env_pair_kernel<N> (given velocities by
    construction); env_step_quad_kernel<NT, VIS, SPLIT>; env_rollout_quad_kernel<NT, VIS, UNI, SPLIT>.  Network
    kernels (sarl_value_kernel, sgan_*_kernel) have no mode."""
    import re
    m = re.search(r"mcn::(\w+)(?:<([^>]*)>)?", name)
    if not m:
        return None
    fam = m.group(1)
    args = [x.strip() for x in (m.group(2) or "").split(",") if x.strip()]
    if fam == "env_pair_kernel":
        return fam, int(args[0]), "given", 1
    if fam == "env_step_kernel":
        return fam, (int(args[1]) or run_humans), _MODES[int(args[3])], 1
    if fam == "env_step_quad_kernel":
        return fam, int(args[0]), "orca", 8 if args[2] == "true" else 4
    if fam == "env_rollout_quad_kernel":
        return fam, int(args[0]), "orca", 8 if args[3] == "true" else 4
    if fam == "env_step_loop_kernel":                 # <NT, VIS>: the one-wavefront ORCA step run T times in one launch
        return fam, int(args[0]), "orca", 1
    return fam, run_humans, None, None


def expected_kernel(E, N, given, steps_per_launch=1):
    """The kernel family the library's dispatcher picks for this launch (csrc/mcn_api.hip fill_step_params,
    env_step.hip launch_env_step, automatic tuning)."""
    waves = -(-E // (64 // N))
    if steps_per_launch > 1:
        if N - 1 <= 4:
            return "env_rollout_quad_kernel"
        # 6-10 humans: mcn_env_rollout runs the one-wavefront step kernel T times inside one launch while the batch is
        # latency-bound (env_step.hip launch_env_step_loop), else T launches of the step kernel
        return "env_step_loop_kernel" if (N <= 10 and waves <= 3072) else "env_step_kernel"
    if given:
        return "env_pair_kernel" if (waves > 4096 and N in (5, 10)) else "env_step_kernel"
    if N - 1 <= 4 and -(-E // (64 // (4 * N))) <= 2800:
        return "env_step_quad_kernel"
    return "env_step_kernel"


def _profiles(pattern):
    """Committed summaries of the LATEST round that has any (profiles/rNN_<pattern>): numbers of an earlier round
    describe another build's kernels and are never mixed in."""
    import glob
    import re
    files = glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + pattern))
    if not files:
        return []
    last = max(re.match(r"r(\d\d)_", os.path.basename(f)).group(1) for f in files)
    out = []
    for f in sorted(files):
        if os.path.basename(f).startswith("r%s_" % last):
            try:
                out.append((os.path.basename(f), json.load(open(f))))
            except Exception:
                pass
    return out


def pmc_traffic(E, given, steps_per_launch=1, N=5):
    """HBM bytes per launch from the committed PMC summaries (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
    passes, gfx950 FETCH_SIZE x2 correction; profiles/rNN_pmc_env_step.json, rNN_pmc_env_rollout.json) of the latest
    round.  A hit must be the SAME kernel family the dispatcher runs for this launch, the same human-policy mode,
    humans, envs and steps per launch; for the rollout kernel -- whose traffic is affine in the steps per launch
    (state once + one action row per step) -- the line through the two nearest measured launch lengths also counts.
    Returns (bytes, source, kernel name) or (None, None, None)."""
    global _PMC
    if _PMC is None:
        _PMC = {}
        for fname, d in _profiles("pmc_env_*.json"):
            for k in d.get("kernels", []):
                c = classify_kernel(k["kernel"], k.get("humans", 5))
                if c is None or c[2] is None:
                    continue
                key = (c[0], c[2], c[1], k["envs"], int(k.get("steps_per_launch", 1)))
                _PMC[key] = (k["traffic_bytes_per_launch"], fname, k["kernel"])
    spl = int(round(steps_per_launch))
    fam, mode = expected_kernel(E, N, given, spl), ("given" if given else "orca")
    hit = _PMC.get((fam, mode, N, E, spl))
    if hit:
        return hit[0], "PMC, %s" % hit[1], hit[2]
    if spl > 1:
        pts = sorted((t, v[0], v[1], v[2]) for (f, m, n, e, t), v in _PMC.items() if (f, m, n, e) == (fam, mode, N, E))
        if len(pts) >= 2:
            pts.sort(key=lambda x: abs(np.log(x[0] / spl)))
            (t0, b0, f0, kn), (t1, b1, _, _) = pts[0], pts[1]
            by = b0 + (b1 - b0) * (spl - t0) / (t1 - t0)
            return int(max(by, 0)), "affine in steps per launch through the PMC runs at %d and %d steps (%s)" % (t0, t1, f0), kn
    return None, None, None


# 256 CUs x 4 SIMDs x 2.4 GHz / (cycles one wave64 vector instruction occupies a SIMD's vector pipe).  The cycle count
# is MEASURED (tools/microbench/valu_issue.hip -> profiles/rNN_valu_issue.txt: independent v_fma_f32 / v_cndmask_b32 /
# v_add_f32 chains with 1, 2 and 4 wavefronts per SIMD); valu_cycles_per_instruction() reads it back.
VALU_CYCLES_DEFAULT = 4.0
_VALU_CYC = None


def valu_cycles_per_instruction():
    """(cycles a float32 wave64 vector instruction holds a SIMD, source): the per-SIMD figure of the v_fma_f32 rows
    of the latest profiles/rNN_valu_issue.txt with the most wavefronts per SIMD -- the rate that more wavefronts
    cannot improve on --, else the default 4 (MI355X_MICROARCH.md issue-cost table)."""
    global _VALU_CYC
    if _VALU_CYC is None:
        import glob
        import re
        _VALU_CYC = (VALU_CYCLES_DEFAULT, "default: MI355X_MICROARCH.md issue-cost table (no profiles/rNN_valu_issue.txt)")
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_valu_issue.txt")))
        if files:
            best = None
            for ln in open(files[-1]):
                m = re.match(r"v_fma_f32 \(VOP3.*?(\d+) wave\(s\)/SIMD:.*?=\s*([0-9.]+) per instruction on the SIMD", ln)
                if m and (best is None or int(m.group(1)) > best[0]):
                    best = (int(m.group(1)), float(m.group(2)))
            if best:
                _VALU_CYC = (best[1], "%s: v_fma_f32, %d wavefronts per SIMD" % (os.path.basename(files[-1]), best[0]))
    return _VALU_CYC


def valu_peak_wave_inst_per_s():
    return 1024 * 2.4e9 / valu_cycles_per_instruction()[0]


_SQ = None


def valu_roofline(E, N, avg_ms, steps_per_launch, rollout, given=False):
    """Instruction-issue roofline of a kernel that HBM does not bound: VALU wave-instructions per env-step from the
    committed SQ_INSTS_VALU pass (profiles/rNN_pmc_sq.json, latest round, same kernel family / mode / humans / envs
    as the dispatcher runs for this launch) x env-steps per launch / the launch duration measured live, against
    1024 SIMDs x 2.4 GHz / the measured cycles per wave64 vector instruction (float64 and transcendental
    instructions occupy the pipe longer, so a kernel heavy in those saturates below 1.0)."""
    global _SQ
    if _SQ is None:
        _SQ = {}
        for fname, d in _profiles("pmc_sq*.json"):
            for k in d.get("kernels", []):
                c = classify_kernel(k["kernel"], k.get("humans", 5))
                if c is not None and c[2] is not None:
                    _SQ[(c[0], c[2], c[1], k["envs"])] = (k, fname)
    fam = expected_kernel(E, N, given, 2 if rollout else 1)
    hit = _SQ.get((fam, "given" if given else "orca", N, E))
    if hit is None:
        return None
    k, fname = hit
    per_env_step = k["valu_per_env_step"]
    ach = per_env_step * E * steps_per_launch / (avg_ms * 1e-3)
    peak = valu_peak_wave_inst_per_s()
    return {"bound": "valu-issue", "kernel": k["kernel"], "achieved": round(ach / 1e9, 2), "peak": round(peak / 1e9, 1),
            "unit": "G wave-instructions/s", "frac": round(ach / peak, 4),
            "peak_cycles_per_instruction": valu_cycles_per_instruction()[0],
            "peak_source": valu_cycles_per_instruction()[1],
            "valu_wave_instructions_per_env_step": round(per_env_step, 2),
            "all_wave_instructions_per_env_step": round(k.get("insts_per_env_step", 0.0), 2),
            "issue_cycles_per_valu_instruction": (round(4.0 * k["counters_per_launch"]["SQ_ACTIVE_INST_VALU"] /
                                                        k["counters_per_launch"]["SQ_INSTS_VALU"], 2)
                                                  if k.get("counters_per_launch", {}).get("SQ_ACTIVE_INST_VALU") else None),
            "wave_cycles_waiting_frac": k.get("wait_frac"), "avg_launch_us": round(avg_ms * 1e3, 3),
            "source": "SQ_INSTS_VALU etc. from %s (separate rocprofv3 --pmc pass), duration live" % fname}


def roofline_entry(E, N, avg_ms, extra=None, given=False, steps_per_launch=1):
    """avg_ms: average duration of ONE launch, which advances E envs by `steps_per_launch` steps."""
    by = (pairwise_bytes_per_env_step(N) if given else algorithmic_bytes_per_env_step(N)) * E * steps_per_launch
    ach = by / (avg_ms * 1e-3) / 1e9
    d = {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": round(ach / HBM_PEAK_GBS, 5),
         "traffic": None, "traffic_source": None, "traffic_kernel": None,
         "kernel": "mcn::" + expected_kernel(E, N, given, int(round(steps_per_launch))),
         "envs_per_launch": E, "env_steps_per_launch": int(round(E * steps_per_launch)),
         "algorithmic_bytes_per_launch": int(round(by)),
         "avg_launch_us": round(avg_ms * 1e3, 3)}
    d["traffic"], d["traffic_source"], d["traffic_kernel"] = pmc_traffic(E, given, steps_per_launch, N)
    if extra:
        d.update(extra)
    return d


_NETS = None


def net_traffic(family, E, N):
    """HBM bytes per launch of a network kernel (sarl_value_kernel, sgan_*_kernel) from profiles/rNN_pmc_nets.json of
    the latest round: (bytes, source) or (None, None)."""
    global _NETS
    if _NETS is None:
        _NETS = {}
        for fname, d in _profiles("pmc_nets.json"):
            for k in d.get("kernels", []):
                _NETS[(k["family"], k["envs"], k["humans"])] = (k["traffic_bytes_per_launch"], fname)
    hit = _NETS.get((family, E, N))
    return (hit[0], "PMC, %s" % hit[1]) if hit else (None, None)


MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32 MFMA (= fp32 vector peak)


def _sarl_policy(device, dt):
    from modelcrowdnav_amd import configs
    from modelcrowdnav_amd.policy.sarl import SARL
    torch.manual_seed(0)                       # default-init weights: the reference ships no trained SARL model
    pol = SARL()
    pol.configure(configs.policy_config())
    pol.kinematics = "holonomic"
    pol.set_device(device)
    pol.set_phase("test")
    pol.time_step = dt
    return pol


def _timed(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); fn()
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def extra_configs(device):
    """BASELINE configs[2], [3] and the per-GPU shard of [4], reported next to the headline workload."""
    out = []
    # ---- config 3: 4096 envs x 5 humans, SARL attention value-net robot ----
    E, N = 4096, 5
    env, _ = build_env(E, N, 0, device)
    pol = _sarl_policy(device, env.time_step)

    def step3():
        a, _ = pol.predict_batch(env)
        env.step(a)
    ms_step = _timed(step3, 30)
    ms_net = _timed(lambda: pol.predict_batch(env), 30)
    flop = 81 * (N * 124100 + 67000) * E
    out.append({"config": "4096 envs x 5 humans, SARL attention value-net robot (81-action look-ahead), ORCA humans",
                "ms_per_step": round(ms_step, 4), "env_steps_per_sec": round(E / ms_step * 1e3, 1),
                "roofline": _sarl_roofline(E, N, ms_net)})
    del env
    # ---- config 4: 4096 envs x 10 humans, model-based rollout with the SGAN predictor ----
    gold = os.path.join(ROOT, "tests", "golden", "g6_sgan.npz")
    if os.path.exists(gold):
        from modelcrowdnav_amd import configs
        from modelcrowdnav_amd.envs import VecModelCrowdSim
        from modelcrowdnav_amd.envs.utils.robot import Robot
        from modelcrowdnav_amd.envs import scenarios as S
        from modelcrowdnav_amd.policy.world_model import VecSGANWorld, generator_from_arrays
        N = 10
        cfg = configs.env_config(**{"sim.human_num": N})
        env = VecModelCrowdSim(E, device)
        env.configure(cfg)
        robot = Robot(cfg, "robot")
        pol4 = _sarl_policy(device, env.time_step)
        robot.set_policy(pol4)
        env.set_robot(robot)
        pool = S.scenario_pool(env.spec(), "test", range(500), N, "circle_crossing")
        env.load_scenarios(pool[np.arange(E) % 500])
        env.attach_rollout(gamma=0.9, pool=pool, case_stride=1, first_cases=(np.arange(E) + 1) % 500)
        gen = generator_from_arrays(np.load(gold), "p", device)          # shipped sgan-p-models/zara1_8 weights
        world = VecSGANWorld(gen, E, N, device, time_step=env.time_step, seed=0)
        world.init_constant_velocity(env.hpos, env.hvel)
        env.sim_world = world

        def step4():
            env.prefetch_world()             # SGAN on a side stream: overlaps the value-network look-ahead
            a, _ = pol4.predict_batch(env)
            env.step(a)                      # picks the prefetched velocities up
        ms4 = _timed(step4, 20)
        ms_sarl10 = _timed(lambda: pol4.predict_batch(env), 20)         # the look-ahead alone: what bounds this step
        ms_sgan = _timed(lambda: world(env.hpos), 96)           # three host noise blocks (VecSGANWorld.draw_noise)
        out.append({"config": "4096 envs x 10 humans, model-based rollout: SGAN (pool-net, zara1_8) world model + SARL robot",
                    "ms_per_step": round(ms4, 4), "env_steps_per_sec": round(E / ms4 * 1e3, 1),
                    "sgan_step_ms": round(ms_sgan, 4), "sarl_lookahead_ms": round(ms_sarl10, 4),
                    "roofline_sarl": _sarl_roofline(E, N, ms_sarl10),
                    # ~0.69 MFLOP per pedestrian for the pooling generator at N = 10 as the reference
                    # writes it (8 encoder LSTM steps, N pool-net MLPs 48 -> 512 -> 8 per pedestrian, context MLP,
                    # decoder LSTM step): `achieved` / `frac` follow that definition.  The kernels execute fewer: the
                    # pool-net's first layer is split into a per-partner and a per-pair part and the spatial embeddings
                    # are folded into the layers they feed (sgan_step.hip); `executed_*` counts the MFMAs really issued
                    # (per 16 pedestrians: encoder 576, pool 2 x 1 064, decoder 160; 2 048 FLOP each).
                    "roofline": _sgan_roofline(E, N, ms_sgan)})
        del env
    # ---- config 5's per-GPU shard: 4096 envs x 10 humans, ORCA humans, random robot actions ----
    # (BASELINE configs[4] = 32 768 x 10 over 8 GPUs.  As in the headline, the pre-drawn actions are known up front, so
    # the steps go to the device as mcn_env_rollout launches of T steps: for 6-10 humans that is the one-wavefront step
    # kernel run T times inside one launch, env_step_loop_kernel.  One mcn_env_step launch per step -- what a
    # policy-in-the-loop caller gets -- is reported beside it.)
    E, N, T10 = 4096, 10, 500
    env, _ = build_env(E, N, 0, device)
    acts10 = make_actions(64, E, E, 0, device)
    ms10, _best = time_kernel_events(env, acts10, 200)
    seq10 = make_actions(T10, E, E, 0, device)
    env.rollout(seq10[:50])
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(4):
        env.rollout(seq10)
    ev[1].record()
    torch.cuda.synchronize()
    ms10r = ev[0].elapsed_time(ev[1]) / 4                        # one launch = T10 steps
    out.append({"config": "4096 envs x 10 humans per GPU (the shard of BASELINE's 32 768 x 10 on 8 GPUs), ORCA humans, "
                          "random robot actions, %d steps per mcn_env_rollout launch" % T10,
                "ms_per_step": round(ms10r / T10, 5), "env_steps_per_sec": round(E * T10 / ms10r * 1e3, 1),
                "steps_per_launch": T10,
                "roofline": roofline_entry(E, N, ms10r, steps_per_launch=T10),
                "roofline_valu": valu_roofline(E, N, ms10r, T10, rollout=True),
                "single_step_launch": {"mode": "one mcn_env_step launch per step, hipGraph of 200",
                                       "ms_per_step": round(ms10, 5), "env_steps_per_sec": round(E / ms10 * 1e3, 1),
                                       "roofline": roofline_entry(E, N, ms10),
                                       "roofline_valu": valu_roofline(E, N, ms10, 1, rollout=False)}})
    del env
    # ---- config 5's WHOLE batch on one GPU: 32 768 envs x 10 humans (what a node with one MI355X would run) ----
    E = 32768
    env, _ = build_env(E, N, 0, device)
    acts10 = make_actions(64, E, E, 0, device)
    ms10w, _best = time_kernel_events(env, acts10, 100)
    out.append({"config": "32 768 envs x 10 humans on ONE GPU (BASELINE configs[4]'s whole batch), ORCA humans, random robot "
                          "actions, one mcn_env_step launch per step (throughput-bound: a looped launch loses here, "
                          "64 vs 81 us per step)",
                "ms_per_step": round(ms10w, 5), "env_steps_per_sec": round(E / ms10w * 1e3, 1),
                "roofline": roofline_entry(E, N, ms10w)})
    return out


def sarl_mfma_per_tile(N):
    """v_mfma_f32_16x16x4_f32 instructions sarl_value_kernel issues per 16-pair tile (csrc/sarl_value.hip: output tiles
    x the k-steps of the input tiles that carry anything): per human mlp1 40 + 266, attention 175 + 175 + 25, mlp2's
    first layer 175; per pair the global half of attention.0 175, mlp2's linear last layer 100 (applied once to the
    attention-weighted sum), mlp3 150 + 266 + 175 + 25."""
    return N * (40 + 266 + 175 + 175 + 25 + 175) + (175 + 100 + 150 + 266 + 175 + 25)


MFMA_BF16_PEAK_TFLOPS = 2516.6   # MI355X_MICROARCH.md: ~2.5 PF dense bf16 = 256 CUs x 4 SIMDs x 2.4 GHz x 1 024 FLOP / cycle


def sarl_x3_mfma_per_tile(N):
    """v_mfma_f32_16x16x32_bf16 instructions sarl_value_kernel<true> issues per 16-pair tile (csrc/mfma_chain.hpp:
    dense_flow_x3 -- output tiles x 32-feature input blocks x 6 piece products): per human mlp1 10*1*6 + 7*5*6,
    attention 7*4*6 + 7*4*6 + 1*4*6, mlp2's first layer 7*4*6; per pair the global half of attention.0 7*4*6, mlp2's
    last layer 4*4*6, mlp3 10*3*6 + 7*5*6 + 7*4*6 + 1*4*6."""
    return N * (60 + 210 + 168 + 168 + 24 + 168) + (168 + 96 + 180 + 210 + 168 + 24)


def sarl_uses_x3():
    """Whether the look-ahead runs its layers on the bf16 pipe (mcn_sarl_net.x3 is always packed by policy/sarl.py;
    mcn_tuning.sarl_x3 = 0 / MCN_SARL_X3=0 keeps the float32 MFMA layers)."""
    from modelcrowdnav_amd import _hip
    return _hip.get_tuning().sarl_x3 != 0


def _sarl_roofline(E, N, ms_net):
    """`achieved` / `frac` count the FLOP the kernel EXECUTES against the dense peak of the matrix instruction it issues
    (tile padding included).  Since round 4 that is v_mfma_f32_16x16x32_bf16 on bfloat16 PIECES of float32 operands (six
    piece products per float32 product: float32-accurate), priced against the bf16 peak; what the same
    time means in float32 terms -- the float32 MFMA instructions the previous kernel issued for the same result, and the
    reference formulation's count (81 x (N x 124 100 + 67 000) per env-step) -- is reported beside it against
    the float32 MFMA peak (both can exceed 1: the work no longer runs on that pipe)."""
    reference = 81 * (N * 124100 + 67000) * E
    tiles = (E * 81 + 15) // 16
    f32_equiv = tiles * sarl_mfma_per_tile(N) * 2048
    tr, src = net_traffic("sarl_value_kernel", E, N)
    if sarl_uses_x3():
        executed, peak = tiles * sarl_x3_mfma_per_tile(N) * 16384, MFMA_BF16_PEAK_TFLOPS
        dtype = "f32 operands as 3 x bf16 pieces, f32 accumulate (v_mfma_f32_16x16x32_bf16, 6 piece products)"
    else:
        executed, peak, dtype = f32_equiv, MFMA_F32_PEAK_TFLOPS, "f32 (v_mfma_f32_16x16x4_f32)"
    return {"bound": "mfma", "kernel": "mcn::sarl_value_kernel", "achieved": round(executed / ms_net / 1e9, 2),
            "peak": peak, "unit": "TFLOP/s", "frac": round(executed / ms_net / 1e9 / peak, 4),
            "traffic": tr, "traffic_source": src, "executed_flop_per_launch": executed,
            "f32_mfma_equivalent_flop_per_launch": f32_equiv,
            "f32_mfma_equivalent_rate": round(f32_equiv / ms_net / 1e9, 2),
            "f32_mfma_equivalent_rate_over_f32_peak": round(f32_equiv / ms_net / 1e9 / MFMA_F32_PEAK_TFLOPS, 4),
            "reference_flop_per_launch": reference, "reference_flop_rate": round(reference / ms_net / 1e9, 2),
            "reference_flop_rate_over_peak": round(reference / ms_net / 1e9 / MFMA_F32_PEAK_TFLOPS, 4),
            "avg_launch_us": round(ms_net * 1e3, 1), "dtype": dtype}


def _sgan_roofline(E, N, ms):
    """`achieved` / `frac` count the FLOP the kernels EXECUTE (MFMAs issued x 2 048); the reference-formulation count
    (~0.69 MFLOP per pedestrian at N = 10, more than the kernels need after the pool-net layer split and
    the folded embeddings) is reported beside it under reference_flop_* and is not a roofline fraction."""
    reference = 0.69e6 * E * N
    tiles = (E * N + 15) // 16
    executed = tiles * (576 + ((N + 4) // 5) * 1064 + 160) * 2048
    parts = [net_traffic(f, E, N) for f in ("sgan_encode_kernel", "sgan_pool_kernel", "sgan_decode_kernel")]
    traffic = sum(p[0] for p in parts) if all(p[0] is not None for p in parts) else None
    return {"bound": "mfma", "kernel": "mcn::sgan_encode_kernel + mcn::sgan_pool_kernel + mcn::sgan_decode_kernel",
            "achieved": round(executed / ms / 1e9, 2), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(executed / ms / 1e9 / MFMA_F32_PEAK_TFLOPS, 4),
            "traffic": traffic, "traffic_source": parts[0][1] if traffic is not None else None,
            "executed_flop_per_launch": int(executed),
            "reference_flop_per_launch": int(reference),
            "reference_flop_rate": round(reference / ms / 1e9, 2),
            "reference_flop_rate_over_peak": round(reference / ms / 1e9 / MFMA_F32_PEAK_TFLOPS, 4),
            "avg_launch_us": round(ms * 1e3, 1), "dtype": "f32 (v_mfma_f32_16x16x4_f32)"}


# the reference's Python CrowdSim.step, `linear` humans, measured once in the build container
REFERENCE_PYTHON_STEPS_PER_S = 8084.0


def cpu_baseline(N, seconds):
    """The C oracle's env step (oracle/mcn_oracle.c) on a bounded sample of the same workload: 4096 envs x 5
    humans from the same scenarios, random table actions.  `value` is ONE thread (the reference is single-threaded);
    `all_cores` runs independent replicas of it as child processes (numpy + ctypes only, they never touch the GPU) on
    the host cores this process may use, started together."""
    import subprocess
    import tempfile
    from oracle import cport, cpu_replica
    from modelcrowdnav_amd.envs import scenarios as S
    cport.lib()                                             # build / load once
    E = 4096
    pool = S.scenario_pool(S.ScenarioSpec(), "test", range(500), N, "circle_crossing")
    sc, tab = pool[np.arange(E) % 500], action_table().numpy()
    n_env_steps, el, steps = cpu_replica.run(cpu_replica.setup(sc, tab, 0), seconds)
    # the cores this process may run on (the GPU box gives one GPU's share of the host), not the host's total
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, 64))                          # bound the number of replica processes
    all_cores = None
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "workload.npz")
        np.savez(path, sc=sc, tab=tab)
        start_at = time.time() + 4.0
        secs = max(2.0, seconds / 3)
        procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_replica", path, str(secs), str(i + 1), str(start_at)],
                                  cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
                 for i in range(cores)]
        outs = []
        for pr in procs:
            try:
                o, _ = pr.communicate(timeout=secs + 60)
                outs.append([float(x) for x in o.split()])
            except Exception:
                pr.kill()
        good = [o for o in outs if len(o) == 3]
        if good:
            # replicas start together and run for the same duration: the aggregate is the sum of their rates
            all_cores = {"value": round(sum(o[0] / o[1] for o in good), 1), "unit": "env-steps/sec", "cores": len(good),
                         "sample": "%d independent replica processes of the same workload, %.1f s each, started "
                                   "together" % (len(good), secs)}
        # a scalar object-per-agent Python step in the shape of the reference's CrowdSim.step
        # (tools/py_scalar_step.py: Python objects and loops, one native ORCA solve per human -- the oracle's solver in
        # place of the absent rvo2), one env at a time in a child process that imports neither torch nor the HIP library
        python_scalar = None
        try:
            psecs = max(2.0, min(10.0, seconds / 2))
            o = subprocess.run([sys.executable, "-m", "tools.py_scalar_step", path, str(psecs), "16"], cwd=ROOT,
                               capture_output=True, text=True, timeout=psecs + 60).stdout.split()
            python_scalar = {"value": round(float(o[0]) / float(o[1]), 1), "unit": "env-steps/sec", "cores": 1,
                             "kind": "python-scalar",
                             "sample": "16 envs x %d humans stepped one env at a time for %.1f s, auto-reset; Python "
                                       "object-per-agent step (the structure of crowd_sim.py:331-434) over the C "
                                       "oracle's ORCA solve, 1 thread" % (N, float(o[1]))}
        except Exception as ex:                             # the baseline is a report, never a reason to lose the line
            python_scalar = {"value": None, "error": repr(ex)}
    return {"value": round(n_env_steps / el, 1), "unit": "env-steps/sec", "cores": 1, "kind": "port",
            "sample": "%d envs x %d humans x %d steps (%.1f s), C oracle, 1 thread; this process may use %d of the "
                      "host's %d cores" % (E, N, steps, el, cores, os.cpu_count()),
            "all_cores": all_cores,
            "python_scalar": python_scalar,
            # the real reference cannot travel to the GPU box (and its ORCA needs rvo2, absent everywhere): the figure
            # SURVEY.md section 6 measured for it in the build container, quoted, not re-measured
            "reference_python": {"value": REFERENCE_PYTHON_STEPS_PER_S, "unit": "env-steps/sec", "cores": 1,
                                 "kind": "quoted-constant",
                                 "sample": "the reference's own CrowdSim.step (crowd_sim.py:331-434), 5 `linear` humans "
                                           "(no ORCA: rvo2 is absent), zero robot action, 123.7 us per step on 1 of 8 "
                                           "Xeon 2.1 GHz cores of the build container; source SURVEY.md section 6"}}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # MCN_BENCH_BACKEND=gloo rehearses the N > 1 control flow on a box with fewer GPUs than ranks (collectives on
    # CPU copies, ranks share devices); the driver's runs use the default: nccl (= RCCL), one GPU per rank
    backend = os.environ.get("MCN_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    cdev = device if backend == "nccl" else torch.device("cpu")        # where collective buffers live
    E, N, K, W = args.envs, args.humans, args.steps, args.warmup
    E_total = E * world
    env, pool = build_env(E, N, rank * E, device)
    acts = make_actions(W + K, E, E_total, rank * E, device)

    # code objects load at a kernel's first launch: do that on a throw-away env, outside any capture, so that a
    # run with --warmup 0 still captures cleanly
    prime, _ = build_env(64, N, 0, device)
    prime.step(acts[0][:64].contiguous())
    prime.rollout(acts[:2, :64].contiguous())
    torch.cuda.synchronize()
    del prime

    # warm-up: W eager steps (also primes the allocator before capture)
    for t in range(W if args.steps_per_launch <= 1 else min(W, 1)):
        env.step(acts[t])
    if args.steps_per_launch > 1 and W > 1:
        env.rollout(acts[1:W])
    torch.cuda.synchronize()

    # the robot's actions are a pre-drawn random sequence, so S consecutive steps go to the device as one
    # mcn_env_rollout call (state stays in registers between steps); S = 1 is one mcn_env_step launch per step
    S = max(1, min(args.steps_per_launch, K))
    chunks = [(t, min(S, K - t)) for t in range(0, K, S)]

    def one_pass():
        """The K timed steps."""
        if S == 1:
            for t in range(K):
                env.step(acts[W + t])
        else:
            for t, n in chunks:
                env.rollout(acts[W + t:W + t + n])

    # untimed probe pass (an extra warm-up of K steps): how many passes fill the minimum timed region?
    ev_s, ev_e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    one_pass()
    torch.cuda.synchronize()
    ev_s.record()
    one_pass()
    ev_e.record()
    torch.cuda.synchronize()
    probe_ms = ev_s.elapsed_time(ev_e)
    R = args.replays if args.replays > 0 else int(min(20000, max(1, np.ceil(args.min_timed_ms / max(probe_ms, 1e-3)))))

    # The pre-drawn action sequence of the whole timed region (R passes of the K steps) is known up front, so
    # consecutive passes go to the device as ONE mcn_env_rollout launch of up to --steps-per-launch steps (the env state
    # stays in registers across them; results equal the separate launches bit for bit,
    # tests/test_env_step_gpu.py::test_rollout_launch_equals_single_steps).  A launch ends with its slowest env group,
    # so 20-step launches run at 3.7 us per step and 1000-step ones at 2.8: SURVEY 8(d) defines the metric over >= 200
    # steady-state steps, and `short_launch` below reports the 20-step figure next to it.
    merge = 1
    if S > 1 and len(chunks) == 1 and R > 1 and not args.no_merge:
        merge = int(max(1, min(R, args.steps_per_launch // K)))
    acts_rep = acts[W:W + K].repeat(merge, 1, 1).contiguous() if merge > 1 else None
    if merge > 1 and args.replays <= 0:
        # size the timed region with the launches it will really use: a whole number of merged launches
        env.rollout(acts_rep)
        torch.cuda.synchronize()
        ev_s.record()
        env.rollout(acts_rep)
        ev_e.record()
        torch.cuda.synchronize()
        per_pass_ms = ev_s.elapsed_time(ev_e) / merge
        R = int(np.ceil(max(1.0, args.min_timed_ms / max(per_pass_ms, 1e-4)) / merge)) * merge
    n_full, n_rem = (R // merge, R % merge) if merge > 1 else (R, 0)

    def run_timed_steps():
        """R back-to-back passes of the K steps."""
        if merge > 1:
            for _ in range(n_full):
                env.rollout(acts_rep)
            if n_rem:
                env.rollout(acts_rep[:n_rem * K])
        else:
            for _ in range(R):
                one_pass()

    use_graph = not args.no_graph
    if use_graph:
        # captured BEFORE the process group exists: ranks are independent until the final gather, and an
        # RCCL watchdog thread polling events during stream capture is a known way to break a capture
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            run_timed_steps()

    gathered = None
    collective = world > 1 or args.force_collective
    if collective:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)
        gathered = torch.empty(world * E, 3, dtype=torch.float32, device=cdev)
        # warm the communicator with the exact collective used later (lazy channel setup stays out of the timing)
        dist.all_gather_into_tensor(gathered, torch.zeros(E, 3, dtype=torch.float32, device=cdev))

    def barrier():
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    ev_s.record()
    if use_graph:
        graph.replay()
    else:
        run_timed_steps()
    ev_e.record()
    if collective:
        # the path's one exchange: episode returns + outcome codes + counts, one fused buffer, one collective
        rb = env.rollout_buffers
        packed = torch.stack([rb["fin_return"][0], rb["fin_info"][0].double(), rb["fin_count"].double()], 1).float()
        dist.all_gather_into_tensor(gathered, packed.to(cdev))
    barrier()
    elapsed = time.perf_counter() - t0
    env_steps_total = float(E * K * R)
    if collective:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        # each rank chose its own pass count R from its own probe: the job's work is the sum over ranks
        ws = torch.tensor([env_steps_total], dtype=torch.float64, device=cdev)
        dist.all_reduce(ws, op=dist.ReduceOp.SUM)
        env_steps_total = float(ws.item())
    # dominant-kernel duration: HIP events (launch stream) bracketing the back-to-back launches of the timed region
    n_launch = (K if S == 1 else len(chunks)) * R if merge == 1 else n_full + (1 if n_rem else 0)
    kernel_ms = ev_s.elapsed_time(ev_e) / n_launch

    rb = env.rollout_buffers
    episodes = int(rb["fin_count"].sum().item())
    mean_ret = float(rb["fin_return"][0][rb["fin_count"] > 0].mean().item()) if episodes else float("nan")

    roof = roofline_entry(E, N, kernel_ms, {"timing": "HIP events around the %d launches of the timed region (%s)" % (
        n_launch, "one hipGraph" if use_graph else "eager")}, steps_per_launch=K * R / n_launch)
    S_eff = S * merge
    if S > 1:
        # state lives in registers across the steps of a launch: HBM sees the state once per launch, not per step
        roof["note"] = ("algorithmic bytes = SURVEY 8(d) per-env-step figure x env-steps per launch; a launch keeps the "
                        "env state in registers for its %d steps, so real HBM traffic (`traffic`) is a fraction of "
                        "that and the kernel is instruction-issue / latency-bound, not HBM-bound: see roofline_valu" % S_eff)
    valu = valu_roofline(E, N, kernel_ms, K * R / n_launch, rollout=S > 1)

    result = {
        "metric": "env-steps/sec (whole node), 5-human CrowdSim x batched envs",
        "value": round(env_steps_total / elapsed, 1), "unit": "env-steps/sec",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": round(elapsed / (env_steps_total / E_total) * 1e3, 6),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "%d envs x %d humans per GPU, ORCA humans, random robot actions (81-entry table), "
                               "auto-reset, %s" % (E, N, "1 mcn_env_step launch per step" if S == 1 else
                                                   "%d steps per mcn_env_rollout launch" % (S * merge)),
                   "steps_per_launch": S * merge, "passes_per_launch": merge, "replays": R, "timed_steps": K * R,
                   "timed_region_ms": round(elapsed * 1e3, 3), "probe_pass_ms": round(probe_ms, 4),
                   "envs_per_gpu": E, "humans": N, "launch": "hipGraph" if use_graph else "eager",
                   "parallelism": "env-shard x%d, no per-step collective" % world},
        "episodes_finished": episodes, "mean_discounted_return": round(mean_ret, 6),
        "gathered_episode_records": None if gathered is None else int((gathered[:, 2] > 0).sum().item()),
        "collective": (dist.get_backend() if collective else None),
        "roofline": roof,
        "roofline_valu": valu,
    }

    if rank == 0 and world == 1 and merge > 1:
        # the same K steps as ONE launch per pass (what a caller with only K actions in hand pays: the launch ends with
        # its slowest env group)
        s_e, e_e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        one_pass()
        torch.cuda.synchronize()
        s_e.record()
        for _ in range(reps):
            one_pass()
        e_e.record()
        torch.cuda.synchronize()
        sl_ms = s_e.elapsed_time(e_e) / reps
        result["short_launch"] = roofline_entry(E, N, sl_ms, {
            "mode": "%d steps per mcn_env_rollout launch, %d back-to-back launches" % (K, reps),
            "env_steps_per_sec": round(E * K / (sl_ms * 1e-3), 1), "us_per_step": round(sl_ms * 1e3 / K, 4),
            "roofline_valu": valu_roofline(E, N, sl_ms, K, rollout=True)}, steps_per_launch=K)
    if rank == 0 and world == 1 and S > 1:
        # the same workload stepped one mcn_env_step launch at a time (policy-in-the-loop callers pay this)
        one_ms, _ = time_kernel_events(env, acts, 200)
        result["single_step_launch"] = roofline_entry(E, N, one_ms, {
            "mode": "one mcn_env_step launch per step, hipGraph of 200", "env_steps_per_sec": round(E / (one_ms * 1e-3), 1),
            "roofline_valu": valu_roofline(E, N, one_ms, 1, rollout=False)})

    if rank == 0 and world == 1 and not args.no_sweep:
        sweep = []
        del env
        for Es in [int(x) for x in args.sweep.split(",") if x]:
            torch.cuda.empty_cache()
            env_s, _ = build_env(Es, N, 0, device)
            a_s = make_actions(8, Es, Es, 0, device)
            for t in range(8):
                env_s.step(a_s[t])
            torch.cuda.synchronize()
            a_ms, _ = time_kernel_events(env_s, a_s, 50)
            sweep.append(roofline_entry(Es, N, a_ms, {"mode": "fused ORCA + pairwise + reward + integrate",
                                                       "env_steps_per_sec": round(Es / (a_ms * 1e-3), 1),
                                                       "roofline_valu": valu_roofline(Es, N, a_ms, 1, rollout=False)}))
            gv = torch.rand(Es, N, 2, dtype=torch.float64, device=device) - 0.5
            env_s.count_hh = False          # ModelCrowdSim.step has no human-human check (model_crowd_sim.py:347-441)
            g_ms, _ = time_kernel_events(env_s, a_s, 50, given_v=gv)
            # the Explorer record (32 B read + 32 B written per env-step) is real traffic the SURVEY 8(d) figure leaves out
            rec_gbs = (pairwise_bytes_per_env_step(N) + 64) * Es / (g_ms * 1e-3) / 1e9
            sweep.append(roofline_entry(Es, N, g_ms, {"mode": "pairwise + reward + integrate (given velocities, "
                                                              "ModelCrowdSim.step)",
                                                       "achieved_incl_explorer_record": round(rec_gbs, 2),
                                                       "frac_incl_explorer_record": round(rec_gbs / HBM_PEAK_GBS, 5),
                                                       "env_steps_per_sec": round(Es / (g_ms * 1e-3), 1),
                                                       "roofline_valu": valu_roofline(Es, N, g_ms, 1, rollout=False, given=True)},
                                        given=True))
            env_s.detach_rollout()          # SURVEY 8(d) "pairwise kernel alone": no Explorer record, no restart
            n_ms, _ = time_kernel_events(env_s, a_s, 50, given_v=gv)
            sweep.append(roofline_entry(Es, N, n_ms, {"mode": "pairwise + reward + integrate, no Explorer record / "
                                                              "auto-restart (plain ModelCrowdSim.step)",
                                                       "env_steps_per_sec": round(Es / (n_ms * 1e-3), 1)}, given=True))
            sweep[-1]["traffic"] = sweep[-1]["traffic_source"] = sweep[-1]["traffic_kernel"] = None   # PMC runs carry the record
            del gv
            del env_s, a_s
        result["roofline_sweep"] = sweep

    if rank == 0 and world == 1 and not args.no_extra:
        torch.cuda.empty_cache()
        result["extra_configs"] = extra_configs(device)

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(N, args.cpu_seconds)
    elif rank == 0:
        result["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(result))
    if collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
