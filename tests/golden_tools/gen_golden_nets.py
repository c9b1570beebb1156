"""Golden fixtures for the learned parts (run through tests/golden_tools/gen_golden.py):

  g5_sarl     CADRL.rotate in/out, SARL ValueNetwork forward with seeded default-init weights,
              MultiHumanRL.predict action_values + chosen action      (cadrl.py:217-252, sarl.py:28-65,
                                                                       multi_human_rl.py:11-63)
  g6_sgan     TrajectoryGenerator.forward on the shipped zara1_8 checkpoints of both families with
              explicit user_noise; checkpoint tensors re-serialised as plain arrays
                                                                      (sgan/models.py:501-553)
  g7_episode  hand-driven reset/act/step episodes: reference CrowdSim + SARL robot
                                                                      (explorer.py:54-125)
  g9_realdata GetRealData on a synthetic TrajNet++ ndjson (tests/golden/g9_scenes.ndjson, written by this tool):
              per-frame observation lists, start_ends, world-model pairs, SGAN cache files, for the default options
              and for windowed scenes with 'moving' / 'stay' padding           (misc.py:47-187, reader.py:44-166)
  g11_queryenv SARL with query_env = true driving the reference CrowdSim: per-step action values through
              env.onestep_lookahead                                            (multi_human_rl.py:35-55, crowd_sim.py:325-329)
  g12_update_memory Explorer.update_memory in RL and imitation-learning mode on synthetic episodes: stored states
              and value targets                                                (explorer.py:153-186)
  g10_trainer Trainer.optimize_batch / optimize_epoch (SGD momentum 0.9, MSE) on a seeded ValueNetwork and memory:
              weights before / after, losses                                   (trainer.py:19-82)
  g13_world   MlpWorld (eval mode) and AttentionWorld forward on seeded scenes with seeded default-init weights
                                                                      (world_model.py:22-106)
  g14_model_env ModelCrowdSim's own host logic: reset with its generators (initial velocities, no reseed), reset(-1),
              set_current_state, and hand-driven episodes whose humans are moved by an MlpWorld module through
              step(new_v=None) / onestep_lookahead                     (model_crowd_sim.py:94-232,268-345,398-441)
  g15_explorer Explorer.run_k_episodes itself (explorer.py:36-151): imitation-learning collection with the ORCA robot,
              a greedy SARL robot, the stay / raw_memory / rawob / cacheFile data-collection mode, counts and rates
  g16_orca_robot BASELINE config 1 as the reference runs it (test.py --policy orca): CrowdSim with an ORCA robot, per-step
              actions / rewards / states, then get_human_times() to the end          (test.py:64-109, crowd_sim.py:219-258)
  g17_sarl_unicycle MultiHumanRL.predict with a unicycle robot ((v, r) actions, heading-dependent propagate, the theta
              feature): action_values + chosen action                  (cadrl.py:82-129,217-252, multi_human_rl.py:11-63)
  g18_lookahead_in_sim CrowdSim with look_ahead_in_sim = true: SARL (query_env) evaluates every action through
              env.onestep_lookahead -> step_in_sim, whose humans are moved by an MlpWorld module
                                                                      (crowd_sim.py:325-329,633-696, multi_human_rl.py:37-38)
  g19_sganworld the E = 1 SGANWorld callable (world_model.py:134-268) on a cache file: rolling 8-frame history, positions
              rounded to 1e-4, late pedestrians padded, generator noise from torch's global stream (seeded per call)
  g20_trainer_sim Trainer_Sim.optimize_epoch (Adam, MSE, 80 / 20 split after random.shuffle, early stopping, best weights
              restored, model.mse) on a seeded AttentionWorld with one batch per epoch    (trainer_sim.py:26-110)
  g21_epsilon MultiHumanRL.predict in the train phase with epsilon 0.5: numpy's global stream decides exploration and the random
              action (multi_human_rl.py:27-29), `last_state` is kept (:60-61)
  g8_datagen  DataGen.gen_data_from_explore_in_mix on a synthetic recorded set: replay-then-freeze and
              replay-then-imagine (MlpWorld) samples, memory contents in IL and RL mode
                                                                      (datagen.py:379-543)
"""
import os

import numpy as np
import torch

from tests.golden_tools import gen_golden as G

OUT, REF = G.OUT, G.REF


def _sarl_policy(seed, N=5):
    from crowd_nav.policy.sarl import SARL
    torch.manual_seed(seed)
    p = SARL()
    p.configure(G.policy_config())
    p.set_device(torch.device("cpu"))
    p.set_phase("test")
    p.time_step = 0.25
    return p


def _state_dict_arrays(model, prefix):
    return {prefix + k.replace(".", "__"): v.detach().cpu().numpy().copy() for k, v in model.state_dict().items()}


def g5_sarl():
    from crowd_sim.envs.utils.state import FullState, ObservableState, JointState
    rng = np.random.RandomState(5)
    rec = {}
    # ---- rotate ----
    pol = _sarl_policy(0)
    for kin in ("holonomic", "unicycle"):
        pol.kinematics = kin
        x = rng.uniform(-4, 4, (512, 14)).astype(np.float32)
        x[:, 4] = rng.uniform(0.3, 0.5, 512); x[:, 13] = rng.uniform(0.3, 0.5, 512); x[:, 7] = rng.uniform(0.5, 1.5, 512)
        y = pol.rotate(torch.from_numpy(x)).numpy()
        rec["rotate_in_" + kin] = x; rec["rotate_out_" + kin] = y
    pol.kinematics = "holonomic"
    # ---- value network forward, seeded default init ----
    for seed in (0, 1):
        p = _sarl_policy(seed)
        rec.update(_state_dict_arrays(p.model, "w%d__" % seed))
        for N in (5, 10, 1):
            x = rng.uniform(-2, 2, (64, N, 13)).astype(np.float32)
            x[:, :, 0] = np.abs(x[:, :, 0]); x[:, :, 2] = 0
            x[:, :, :6] = x[:, :1, :6]            # self part identical across humans, as transform() builds it
            with torch.no_grad():
                v = p.model(torch.from_numpy(x)).numpy()
                att = []
                for b in range(x.shape[0]):        # reference only keeps weights[0]; run per sample
                    p.model(torch.from_numpy(x[b:b + 1]))
                    att.append(p.model.attention_weights.copy())
            rec["vn%d_in_N%d" % (seed, N)] = x
            rec["vn%d_out_N%d" % (seed, N)] = v
            rec["vn%d_att_N%d" % (seed, N)] = np.array(att)
    # ---- predict: action_values + chosen action ----
    for seed in (0, 1):
        for N in (5, 10):
            p = _sarl_policy(seed)
            p.kinematics = "holonomic"
            states, vals, acts = [], [], []
            for s in range(48):
                rpx, rpy = rng.uniform(-3, 3, 2)
                near_goal = s % 6 == 5
                gx, gy = (rpx + rng.uniform(-0.2, 0.2), rpy + rng.uniform(-0.2, 0.2)) if near_goal else rng.uniform(-4, 4, 2)
                me = FullState(rpx, rpy, rng.uniform(-1, 1), rng.uniform(-1, 1), 0.3, gx, gy, 1.0, np.pi / 2)
                hs = []
                for i in range(N):
                    if s % 3 == 0 and i < 2:
                        a, d = rng.uniform(0, 2 * np.pi), rng.uniform(0.5, 1.3)
                        hx, hy = rpx + d * np.cos(a), rpy + d * np.sin(a)
                    else:
                        hx, hy = rng.uniform(-4, 4, 2)
                    hs.append(ObservableState(hx, hy, rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(0.3, 0.5)))
                js = JointState(me, hs)
                with torch.no_grad():
                    act = p.predict(js)
                row = [me.px, me.py, me.vx, me.vy, me.radius, me.gx, me.gy, me.v_pref, me.theta]
                states.append((np.array(row), np.array([[h.px, h.py, h.vx, h.vy, h.radius] for h in hs])))
                acts.append([act.vx, act.vy])
                vals.append(np.array(p.action_values) if not p.reach_destination(js) else np.full(81, np.nan))
            rec["pred%d_N%d_self" % (seed, N)] = np.array([s[0] for s in states])
            rec["pred%d_N%d_humans" % (seed, N)] = np.array([s[1] for s in states])
            rec["pred%d_N%d_values" % (seed, N)] = np.array(vals)
            rec["pred%d_N%d_action" % (seed, N)] = np.array(acts)
            rec["pred%d_N%d_table" % (seed, N)] = np.array([[a.vx, a.vy] for a in p.action_space])
    np.savez_compressed(os.path.join(OUT, "g5_sarl.npz"), **rec)
    print("g5_sarl: %d arrays" % len(rec))


def g7_episode():
    """Reference CrowdSim (ORCA humans through the rvo2 stand-in) + SARL robot with seeded weights, driven by
    the loop of Explorer.run_k_episodes (explorer.py:54-70,124): per-step chosen action, reward, done, info,
    discounted return."""
    rec = {}
    for seed, N in ((0, 5), (1, 5), (0, 10)):
        torch.manual_seed(seed)
        env, robot, pol = G.make_env("CrowdSim", robot_policy="sarl", humans_policy="orca", human_num=N)
        rec.update(_state_dict_arrays(pol.model, "ep%d_N%d_w__" % (seed, N)))
        for case in (0, 1, 2, 3):
            ob = env.reset("test", case)
            done, t = False, 0
            A, R, D, I, S = [], [], [], [], []
            while not done:
                rob, hum = G.full_state_rows(env)
                with torch.no_grad():
                    action = robot.act(ob)
                ob, reward, done, info = env.step(action)
                A.append([action.vx, action.vy]); R.append(reward); D.append(done); I.append(G.info_code(info))
                S.append(np.concatenate([rob, hum.ravel()]))
                t += 1
            ret = sum([pow(0.9, k * robot.time_step * robot.v_pref) * r for k, r in enumerate(R)])
            key = "ep%d_N%d_c%d_" % (seed, N, case)
            rec[key + "actions"] = np.array(A); rec[key + "rewards"] = np.array(R); rec[key + "done"] = np.array(D)
            rec[key + "info"] = np.array(I); rec[key + "states"] = np.array(S); rec[key + "return"] = np.array(ret)
            rec[key + "time"] = np.array(env.global_time)
            print("  episode seed %d N %d case %d: %d steps, outcome %d, return %.4f" % (seed, N, case, t, I[-1], ret))
    np.savez_compressed(os.path.join(OUT, "g7_episode.npz"), **rec)
    print("g7_episode: %d arrays" % len(rec))


def synthetic_recorded_episodes(seed=11, n_epi=7, N=5):
    """Recorded 'real' episodes in the raw_memory layout of misc.py:85-89: per frame an [N,5] row block
    (px, py, vx, vy, radius).  Tracks are straight walks through the robot's corridor with a slow drift; velocities
    are frame differences times the frame rate, as misc.py:GetVel computes them."""
    rng = np.random.RandomState(seed)
    fps = 1.0 / 0.25
    epis = []
    for e in range(n_epi):
        T = int(rng.randint(22, 44))
        a = rng.uniform(0, 2 * np.pi, N)
        start = np.stack([3.5 * np.cos(a), 3.5 * np.sin(a) - 1.0], 1) + rng.uniform(-0.4, 0.4, (N, 2))
        goal = -start + rng.uniform(-0.6, 0.6, (N, 2)) + np.array([0.0, -2.0])
        speed = rng.uniform(0.5, 1.1, N)
        d = goal - start
        step = d / np.linalg.norm(d, axis=1, keepdims=True) * speed[:, None] / fps
        pos = start[None] + step[None] * np.arange(T)[:, None, None]
        pos = pos + 0.05 * np.sin(np.arange(T)[:, None, None] * 0.3 + a[None, :, None])
        pos = np.round(pos, 3)                      # dataset coordinates come with few decimals
        vel = np.zeros_like(pos)
        vel[1:] = (pos[1:] - pos[:-1]) * fps
        epis.append(np.concatenate([pos, vel, np.full((T, N, 1), 0.3)], 2))
    return epis


def g8_datagen():
    """Reference DataGen (datagen.py:379-543) over ModelCrowdSim + SARL robot with seeded weights."""
    import random
    from crowd_nav.utils.datagen import DataGen
    from crowd_nav.utils.memory import ReplayMemory
    from crowd_nav.policy.world_model import MlpWorld
    from crowd_sim.envs.utils.state import ObservableState
    from crowd_sim.envs.utils.info import Nothing
    rec = {}
    epis = synthetic_recorded_episodes()
    for i, e in enumerate(epis):
        rec["epi%d" % i] = e
    raw = []
    for e in epis:
        start_ends = [[e[0, h, 0], e[0, h, 1], e[-1, h, 0], e[-1, h, 1]] for h in range(e.shape[1])]
        for t in range(e.shape[0]):
            ob = [ObservableState(*e[t, h].tolist()) for h in range(e.shape[1])]
            raw.append((ob, 0, t == e.shape[0] - 1, Nothing(), start_ends))
    runs = [("il_freeze", dict(imitation_learning=True, add_sim=False, random_epi=True), 3, 12),
            ("rl_imagine", dict(imitation_learning=False, add_sim=True, random_epi=False), 4, 10),
            ("il_static", dict(imitation_learning=True, add_sim=True, random_epi=True, static_end=9), 5, 9),
            ("il_replace_rand", dict(imitation_learning=True, add_sim=True, random_epi=True, replace_robot=True,
                                     random_robot=True), 6, 8),
            ("rl_replace_long", dict(imitation_learning=False, add_sim=False, random_epi=False, replace_robot=True,
                                     random_robot=False), 7, 7),
            ("il_view3", dict(imitation_learning=True, add_sim=False, random_epi=True, view_human=3), 8, 8),
            ("view_dist", dict(add_sim=True, random_epi=True, view_distance=3.0, updateMemory=False), 9, 8),
            ("view_dist2", dict(add_sim=False, random_epi=False, view_distance=2.5, view_human=2, updateMemory=False),
             10, 7)]
    # recordings whose crowd grows over time (misc.py padding_first='none'): pedestrian k enters at frame enter[k]
    rng = np.random.RandomState(23)
    ragged_raw = []
    for i, e in enumerate(synthetic_recorded_episodes(seed=29, n_epi=6)):
        T, P = e.shape[0], e.shape[1]
        enter = np.sort(np.concatenate([[0, 0], rng.randint(1, T // 2, P - 2)]))
        count = np.array([(enter <= t).sum() for t in range(T)])
        rec["repi%d" % i], rec["repi%d_count" % i] = e, count
        first = [int(enter[h]) for h in range(P)]
        start_ends = [[e[first[h], h, 0], e[first[h], h, 1], e[-1, h, 0], e[-1, h, 1]] for h in range(P)]
        for t in range(T):
            ob = [ObservableState(*e[t, h].tolist()) for h in range(int(count[t]))]
            ragged_raw.append((ob, 0, t == T - 1, Nothing(), start_ends))
    runs += [("il_sgan", dict(imitation_learning=True, add_sim=True, random_epi=True, sgan_world=True), 14, 8),
             ("ragged_eval", dict(add_sim=False, random_epi=False, updateMemory=False), 11, 7),
             ("ragged_view", dict(add_sim=False, random_epi=True, updateMemory=False, view_distance=3.0, view_human=3), 12, 7),
             ("ragged_replace", dict(add_sim=False, random_epi=False, updateMemory=False, replace_robot=True,
                                     random_robot=False), 13, 6),
             ("ragged_sgan", dict(add_sim=True, random_epi=True, updateMemory=False, sgan_world=True), 15, 7)]
    for name, kw, seed, num in runs:
        torch.manual_seed(3)            # default-init SARL weights that happen to drive to the goal: memory gets rows
        env, robot, pol = G.make_env("ModelCrowdSim", robot_policy="sarl", humans_policy="orca", human_num=5)
        torch.manual_seed(100 + seed)
        world = MlpWorld(4 if kw.get("replace_robot") else 5)      # the replaced pedestrian leaves the crowd
        world.eval()
        kw = dict(kw)
        cache_dir = None
        if kw.pop("sgan_world", False):
            # the shipped pooling SGAN generator as the world model (world_model.py:134-268), fed the generator's own
            # `user_noise` argument with zeros so that the run does not depend on torch's random stream
            import tempfile
            from crowd_nav.policy.world_model import SGANWorld

            class ZeroNoise(torch.nn.Module):
                def __init__(self, g):
                    super().__init__()
                    self.g, self.decoder = g, g.decoder

                def forward(self, obs_traj, obs_traj_rel, seq_start_end):
                    return self.g(obs_traj, obs_traj_rel, seq_start_end, user_noise=torch.zeros(seq_start_end.shape[0], 8))
            cache_dir = tempfile.TemporaryDirectory()
            kw["sgan_genfile"] = os.path.join(cache_dir.name, "generate.txt")
            world = SGANWorld(kw["sgan_genfile"], torch.device("cpu"), obs_len=8, time_step=0.25,
                              pretrainPath=os.path.join(REF, "sgan", "models", "sgan-p-models", "zara1_8_model.pt"))
            world.generator = ZeroNoise(world.generator)
        env.sim_world = world
        env.device = torch.device("cpu")
        rec.update(_state_dict_arrays(pol.model, "w__"))          # same seeded weights in every run
        if isinstance(world, MlpWorld):
            rec.update(_state_dict_arrays(world, name + "_world__"))
        memory = ReplayMemory(100000)
        dg = DataGen(memory, robot, env, pol)
        dg.raw_memory = ragged_raw if name.startswith("ragged") else raw
        dg.update_target_model(pol.model)
        random.seed(seed)
        out = dg.gen_data_from_explore_in_mix(num, phase="val", min_end=8, returnRate=False, **kw)
        rec[name + "_out"] = np.array(out, np.float64)
        rec[name + "_states"] = np.stack([m[0].numpy() for m in memory.memory]) if len(memory.memory) else np.zeros((0, 5, 13), np.float32)
        rec[name + "_values"] = np.array([float(m[1].item()) for m in memory.memory], np.float32)
        rec[name + "_counter"] = np.array(dg.counter)
        print("  %s: out %s, %d memory rows" % (name, out, len(memory.memory)))
    np.savez_compressed(os.path.join(OUT, "g8_datagen.npz"), **rec)
    print("g8_datagen: %d arrays" % len(rec))


def write_synthetic_ndjson(path, seed=21):
    """A small TrajNet++-style recording: three groups of overlapping scene entries separated by silent gaps,
    pedestrians that enter and leave at different frames, one track with a missing sample, frames every 2 ticks."""
    import json
    rng = np.random.RandomState(seed)
    lines, sid = [], 0
    ped0 = 0
    for grp, (f0, n_frames, n_ped) in enumerate(((10, 16, 4), (80, 22, 6), (200, 12, 3))):
        frames = [f0 + 2 * i for i in range(n_frames)]
        tracks = {}
        for k in range(n_ped):
            a = int(rng.randint(0, max(1, n_frames // 3))) if k else 0          # pedestrian 0 of a group is there from the start
            b = int(rng.randint(2 * n_frames // 3, n_frames)) if k else n_frames - 1
            p0, v = rng.uniform(-5, 5, 2), rng.uniform(-0.35, 0.35, 2)
            pts = [(frames[i], ped0 + k, round(float(p0[0] + v[0] * (i - a) + 0.03 * np.sin(i)), 2),
                    round(float(p0[1] + v[1] * (i - a) + 0.03 * np.cos(i)), 2)) for i in range(a, b + 1)]
            if grp == 1 and k == 2 and len(pts) > 6:
                del pts[4]                                                       # a dropped detection
            tracks[ped0 + k] = pts
        # scene entries: overlapping windows over the group, each with a primary pedestrian present at its start
        for w in range(0, n_frames - 8, 5):
            s_f, e_f = frames[w], frames[min(w + 9, n_frames - 1)]
            prim = [pid for pid, pts in tracks.items() if any(r[0] == s_f for r in pts)][0]
            lines.append(json.dumps({"scene": {"id": sid, "p": prim, "s": s_f, "e": e_f, "fps": 2.5, "tag": [0, []]}}))
            sid += 1
        for f in frames:
            for pid, pts in tracks.items():
                for r in pts:
                    if r[0] == f:
                        lines.append(json.dumps({"track": {"f": r[0], "p": r[1], "x": r[2], "y": r[3]}}))
        ped0 += n_ped
    with open(path, "w") as fh:
        fh.write("\n".join(lines) + "\n")


def g9_realdata():
    """Reference GetRealData (misc.py:47-116) on the synthetic recording."""
    import tempfile
    from crowd_nav.utils.misc import GetRealData, StoreAction
    path = os.path.join(OUT, "g9_scenes.ndjson")
    write_synthetic_ndjson(path)
    rec = {}
    cases = [("default_test", dict(phase="test")),
             ("default_train", dict(phase="train")),
             ("default_val", dict(phase="val")),
             ("win_moving", dict(phase="test", stride=2, windows_size=6, padding_last="moving", padding_first="stay")),
             ("win_slice", dict(phase="val", stride=3, windows_size=5, dataset_slice=[1, 6]))]
    for name, kw in cases:
        with tempfile.TemporaryDirectory() as td:
            raw, rawob = GetRealData(dataset_file=path, Store_for_world_fn=StoreAction, cacheFile=td, **kw)
            rows = raw.memory
            n = np.array([len(r[0]) for r in rows], np.int64)
            flat = np.array([[h.px, h.py, h.vx, h.vy, h.radius] for r in rows for h in r[0]], np.float64).reshape(-1, 5)
            rec[name + "_count"] = n
            rec[name + "_obs"] = flat
            rec[name + "_done"] = np.array([r[2] for r in rows])
            se = [np.array(r[4], np.float64).reshape(-1, 4) for r in rows]
            rec[name + "_se_count"] = np.array([len(x) for x in se], np.int64)
            rec[name + "_se"] = np.concatenate(se) if se else np.zeros((0, 4))
            pairs = rawob.memory
            rec[name + "_pair_count"] = np.array([p[0].shape[0] for p in pairs], np.int64)
            rec[name + "_pair_cur"] = np.concatenate([p[0].numpy() for p in pairs]) if pairs else np.zeros((0, 4), np.float32)
            rec[name + "_pair_next_count"] = np.array([p[1].shape[0] for p in pairs], np.int64)
            rec[name + "_pair_next"] = np.concatenate([p[1].numpy().reshape(-1, 2) for p in pairs]) if pairs else np.zeros((0, 2), np.float32)
            files = sorted(os.listdir(td), key=lambda x: int(x.split(".")[0]))
            rec[name + "_cache_files"] = np.array(len(files))
            for f in files:
                rec[name + "_cache_" + f.split(".")[0]] = np.frombuffer(open(os.path.join(td, f), "rb").read(), np.uint8)
            print("  %s: %d rows, %d episodes, %d pairs, %d cache files" % (name, len(rows), int(rec[name + "_done"].sum()),
                                                                          len(pairs), len(files)))
    np.savez_compressed(os.path.join(OUT, "g9_realdata.npz"), **rec)
    print("g9_realdata: %d arrays" % len(rec))


def g6_sgan():
    from crowd_nav.policy.world_model import get_generator
    from sgan.utils import relative_to_abs
    rng = np.random.RandomState(6)
    rec = {}
    dev = torch.device("cpu")
    for fam in ("sgan-models", "sgan-p-models"):
        path = os.path.join(REF, "sgan", "models", fam, "zara1_8_model.pt")
        ck = torch.load(path, map_location="cpu", weights_only=True)
        gen = get_generator(ck, dev)
        gen.decoder.seq_len = 1
        tag = "p" if "p-models" in fam else "np"
        for k, v in ck["g_state"].items():
            rec["%s__w__%s" % (tag, k.replace(".", "__"))] = v.numpy().copy()
        args = ck["args"]
        rec[tag + "__pooling_type"] = np.array(str(args["pooling_type"]))
        rec[tag + "__dims"] = np.array([args["embedding_dim"], args["encoder_h_dim_g"], args["decoder_h_dim_g"],
                                        args["mlp_dim"], args["bottleneck_dim"], args["noise_dim"][0]], np.int64)
        for scene_cfg in ((6, 5), (3, 10), (4, 1)):
            S, N = scene_cfg
            pos0 = rng.uniform(-4, 4, (S * N, 2))
            vel = rng.uniform(-0.4, 0.4, (S * N, 2))
            traj = np.stack([pos0 + vel * t + rng.normal(0, 0.02, (S * N, 2)) for t in range(8)], 0)
            traj = np.around(traj, 4).astype(np.float32)                       # [8, S*N, 2]
            rel = np.zeros_like(traj); rel[1:] = traj[1:] - traj[:-1]
            sse = torch.tensor([[i * N, (i + 1) * N] for i in range(S)], dtype=torch.long)
            noise = torch.from_numpy(rng.normal(0, 1, (S, 8)).astype(np.float32))
            with torch.no_grad():
                pr = gen(torch.from_numpy(traj), torch.from_numpy(rel), sse, user_noise=noise)
                pa = relative_to_abs(pr, torch.from_numpy(traj)[-1])
            key = "%s__S%d_N%d__" % (tag, S, N)
            rec[key + "obs_traj"] = traj; rec[key + "obs_rel"] = rel; rec[key + "noise"] = noise.numpy()
            rec[key + "pred_rel"] = pr.numpy(); rec[key + "pred_abs"] = pa.numpy()
    np.savez_compressed(os.path.join(OUT, "g6_sgan.npz"), **rec)
    print("g6_sgan: %d arrays" % len(rec))


def g11_queryenv():
    """Reference CrowdSim (ORCA humans through the rvo2 stand-in) + SARL robot with `query_env = true`
    (multi_human_rl.py:37-38: every candidate action is evaluated through env.onestep_lookahead): episodes with an
    invisible and with a visible robot; per step the chosen action, the 81 action values, reward, info."""
    rec = {}
    for vis in (False, True):
        torch.manual_seed(2)
        env, robot, pol = G.make_env("CrowdSim", robot_policy="sarl", humans_policy="orca", human_num=5, robot_visible=vis)
        pol.query_env = True
        pol.set_env(env)
        tag = "vis%d_" % int(vis)
        if not vis:                          # same seed: one set of weights
            rec.update(_state_dict_arrays(pol.model, "w__"))
        for case in (0, 5):
            ob = env.reset("test", case)
            done, t = False, 0
            A, V, R, I = [], [], [], []
            while not done and t < 25:
                with torch.no_grad():
                    action = robot.act(ob)
                V.append(np.array(pol.action_values, np.float64))
                ob, reward, done, info = env.step(action)
                A.append([action.vx, action.vy]); R.append(reward); I.append(G.info_code(info))
                t += 1
            key = tag + "c%d_" % case
            rec[key + "actions"] = np.array(A); rec[key + "values"] = np.array(V)
            rec[key + "rewards"] = np.array(R); rec[key + "info"] = np.array(I)
            print("  query_env visible %d case %d: %d steps, outcome %d" % (vis, case, t, I[-1]))
    np.savez_compressed(os.path.join(OUT, "g11_queryenv.npz"), **rec)
    print("g11_queryenv: %d arrays" % len(rec))


def g12_update_memory():
    """Reference Explorer.update_memory (explorer.py:153-186) on three synthetic episodes of 7 / 12 / 1 steps: RL mode
    (value = r + gamma_bar * target_model(next state), terminal value = r) with a seeded SARL ValueNetwork as target
    model, and imitation-learning mode (discounted tail sums; states transformed by the target policy)."""
    from crowd_sim.envs.utils.state import FullState, ObservableState, JointState
    from crowd_nav.utils.explorer import Explorer
    from crowd_nav.utils.memory import ReplayMemory
    rng = np.random.RandomState(12)
    pol = _sarl_policy(41)
    rec = dict(_state_dict_arrays(pol.model, "w__"))

    class _Robot(object):
        time_step, v_pref = 0.25, 1.0
    lens = (7, 12, 1)
    for mode in ("rl", "il"):
        for e, L in enumerate(lens):
            rewards = [float(x) for x in rng.uniform(-0.05, 0.0, L)]
            rewards[-1] = 1.0 if e != 1 else -0.25
            mem = ReplayMemory(100)
            ex = Explorer(None, _Robot(), torch.device("cpu"), mem, 0.9, target_policy=pol)
            ex.update_target_model(pol.model)
            if mode == "rl":
                states = [torch.from_numpy(rng.normal(0, 1, (5, 13)).astype(np.float32)) for _ in range(L)]
                with torch.no_grad():
                    ex.update_memory(states, None, rewards, imitation_learning=False)
                rec["rl_e%d_states" % e] = np.stack([s_.numpy() for s_ in states])
            else:
                joint = []
                for _ in range(L):
                    me = FullState(*[float(x) for x in rng.uniform(-3, 3, 4)], 0.3, 0.0, 4.0, 1.0, 0.0)
                    hs = [ObservableState(*[float(x) for x in rng.uniform(-3, 3, 4)], 0.3) for _ in range(5)]
                    joint.append(JointState(me, hs))
                ex.update_memory(joint, None, rewards, imitation_learning=True)
                rec["il_e%d_states" % e] = np.stack([m[0].numpy() for m in mem.memory])
            rec["%s_e%d_rewards" % (mode, e)] = np.array(rewards)
            rec["%s_e%d_values" % (mode, e)] = np.array([float(m[1]) for m in mem.memory], np.float32)
            rec["%s_e%d_mem_states" % (mode, e)] = np.stack([m[0].numpy() for m in mem.memory])
    np.savez_compressed(os.path.join(OUT, "g12_update_memory.npz"), **rec)
    print("g12_update_memory: %d arrays" % len(rec))


def g10_trainer():
    """Reference Trainer (crowd_nav/utils/trainer.py:19-82) on a seeded SARL ValueNetwork and a seeded memory of
    (state [5,13], value [1]) pairs.  optimize_batch: the memory holds exactly one batch, so every step sees all rows
    (the DataLoader's shuffle only permutes them inside the batch); optimize_epoch: two full batches per epoch in the
    order of the DataLoader's permutation, which is recorded."""
    from crowd_sim.envs.utils.state import JointState  # noqa: F401  (first: the reference's packages import each other)
    from crowd_nav.utils.trainer import Trainer
    from crowd_nav.utils.memory import ReplayMemory
    rec = {}
    rng = np.random.RandomState(10)
    for tag, n_rows, bs in (("batch", 100, 100), ("epoch", 200, 100)):
        pol = _sarl_policy(31)
        model = pol.get_model()
        if tag == "batch":                  # same seed: both cases start from these weights
            rec.update(_state_dict_arrays(model, "w0__"))
        states = rng.normal(0, 1, (n_rows, 5, 13)).astype(np.float32)
        values = rng.uniform(-0.5, 1.0, (n_rows, 1)).astype(np.float32)
        mem = ReplayMemory(n_rows)
        for s_, v_ in zip(states, values):
            mem.push((torch.from_numpy(s_), torch.from_numpy(v_)))
        tr = Trainer(model, mem, torch.device("cpu"), bs)
        tr.set_learning_rate(0.01)
        torch.manual_seed(77)
        if tag == "batch":
            losses = [tr.optimize_batch(1) for _ in range(3)]
        else:
            # the DataLoader draws one randperm per epoch from the global generator: record it
            g_state = torch.get_rng_state()
            losses = [tr.optimize_epoch(1) for _ in range(2)]
            torch.set_rng_state(g_state)
            # replay an identical DataLoader over the row indices: it consumes the generator exactly as the Trainer's
            import torch.utils.data as tud

            class _Rows(tud.Dataset):
                def __len__(self):
                    return n_rows

                def __getitem__(self, i):
                    return i
            loader = tud.DataLoader(_Rows(), bs, shuffle=True)
            perms = [np.concatenate([b.numpy() for b in loader]).astype(np.int64) for _ in range(2)]
            rec[tag + "_perms"] = np.stack(perms)
        rec[tag + "_states"], rec[tag + "_values"] = states, values
        rec[tag + "_losses"] = np.array(losses, np.float64)
        rec.update(_state_dict_arrays(model, tag + "_w1__"))
    np.savez_compressed(os.path.join(OUT, "g10_trainer.npz"), **rec)
    print("g10_trainer: %d arrays" % len(rec))


def g13_world():
    """The two world-model modules of crowd_nav/policy/world_model.py:22-106 evaluated by the reference itself: input
    rows are [px, py, vx, vy] per pedestrian (datagen.py:505-510 builds them that way), outputs the next velocities.
    MlpWorld runs in eval mode (its Dropout layers are the identity, as when the training script imagines)."""
    from crowd_sim.envs.utils.state import JointState  # noqa: F401
    from crowd_nav.policy.world_model import MlpWorld, AttentionWorld
    rng = np.random.RandomState(13)
    rec = {}

    def scenes(B, N):
        x = np.concatenate([rng.uniform(-4.5, 4.5, (B, N, 2)), rng.uniform(-1.2, 1.2, (B, N, 2))], axis=2)
        x[::7, :, 2:] = 0.0                                    # standing crowds
        return x.reshape(B, N * 4).astype(np.float32)
    for N in (1, 5, 10):
        torch.manual_seed(130 + N)
        m = MlpWorld(N).eval()
        x = scenes(192, N)
        with torch.no_grad():
            y = m(torch.from_numpy(x)).numpy()
        rec.update(_state_dict_arrays(m, "mlp%d_w__" % N))
        rec["mlp%d_in" % N], rec["mlp%d_out" % N] = x, y
    torch.manual_seed(139)
    a = AttentionWorld().eval()
    rec.update(_state_dict_arrays(a, "attn_w__"))
    for N in (2, 5, 10):
        x = scenes(192, N)
        with torch.no_grad():
            y = a(torch.from_numpy(x)).numpy()
        rec["attn%d_in" % N], rec["attn%d_out" % N] = x, y
        rec["attn%d_weights0" % N] = np.asarray(a.attention_weights, np.float32)        # scene 0's attention weights
    np.savez_compressed(os.path.join(OUT, "g13_world.npz"), **rec)
    print("g13_world: %d arrays" % len(rec))


def g14_model_env():
    """The reference's ModelCrowdSim driven as its callers do (train_model_based_sgan.py, datagen.py:434-476)."""
    from crowd_sim.envs.utils.state import ObservableState, FullState
    from crowd_sim.envs.utils.action import ActionXY
    from crowd_nav.policy.world_model import MlpWorld
    rec, meta, k = {}, [], 0
    # ---- reset: the seeded branch does not reseed numpy (model_crowd_sim.py:296 is commented out): the caller's
    # global generator state decides, consecutive resets continue the stream
    for robot_policy in ("sarl", "orca"):
        for rule in ("circle_crossing", "square_crossing", "mixed"):
            # (10 humans with random radii can jam the reference's rejection loop for ever: 20 positions + goals of up
            # to 1.2 m spacing on a 25 m circle)
            for N, rnd in ((5, False), (10, False), (3, True), (5, True)):
                env, robot, pol = G.make_env("ModelCrowdSim", robot_policy=robot_policy, human_num=N, randomize=rnd,
                                             train_val_sim=rule, test_sim=rule)
                for phase in ("test", "val", "train"):
                    np.random.seed(1400 + k)
                    for rep in range(3):
                        env.human_num = N
                        env.reset(phase, 7 if rep == 0 else None)
                        rob, hum = G.full_state_rows(env)
                        rec["reset_rob_%d" % k], rec["reset_hum_%d" % k] = rob, hum
                        meta.append((robot_policy == "sarl", rule, N, rnd, phase, rep, 1400 + k - rep, hum.shape[0],
                                     env.case_counter[phase]))
                        k += 1
    for j, name in enumerate(("multiagent", "rule", "N", "randomize", "phase", "rep", "seed", "nh", "counter_after")):
        rec["reset_meta_" + name] = np.array([m[j] for m in meta])
    rec["case_size_train"] = np.array(env.case_size["train"], np.int64)
    env, robot, pol = G.make_env("ModelCrowdSim", robot_policy="orca", human_num=5)
    env.reset("test", -1)                                           # the three-human debugging scene
    rec["debug_rob"], rec["debug_hum"] = G.full_state_rows(env)
    # ---- set_current_state
    rng = np.random.RandomState(14)
    env, robot, pol = G.make_env("ModelCrowdSim", robot_policy="orca", human_num=5)
    for c in range(4):
        n = (4, 5, 2, 7)[c]
        rows = np.concatenate([rng.uniform(-4, 4, (n, 2)), rng.uniform(-1, 1, (n, 2)), rng.uniform(0.2, 0.5, (n, 1))], 1)
        obs = [ObservableState(*r) for r in rows]
        info = None if c == 0 else FullState(*rng.uniform(-4, 4, 2), 0.3, -0.2, 0.3, *rng.uniform(-4, 4, 2), 1.0, 0.7)
        env.set_current_state(obs, info, phase=("train", "val", "test", "train")[c])
        rec["scs_obs_%d" % c] = rows
        rec["scs_info_%d" % c] = np.zeros(0) if info is None else np.array([info.px, info.py, info.gx, info.gy])
        rec["scs_rob_%d" % c], rec["scs_hum_%d" % c] = G.full_state_rows(env)
        rec["scs_global_time_%d" % c] = np.array(env.global_time, np.float64)
    # ---- episodes: humans moved by an MlpWorld module (the non-SGAN branch of :398-407), robot actions from the table
    pol0 = _sarl_policy(0)
    pol0.kinematics = "holonomic"
    pol0.build_action_space(1.0)
    acts = np.array([[a.vx, a.vy] for a in pol0.action_space], np.float64)
    for e, (N, rule, seed) in enumerate(((5, "circle_crossing", 3), (5, "square_crossing", 4), (3, "circle_crossing", 5),
                                         (10, "circle_crossing", 6), (2, "square_crossing", 9),
                                         (2, "circle_crossing", 11))):
        env, robot, pol = G.make_env("ModelCrowdSim", robot_policy="orca", human_num=N, train_val_sim=rule, test_sim=rule)
        env.device = torch.device("cpu")
        torch.manual_seed(140 + e)
        world = MlpWorld(N).eval()
        with torch.no_grad():
            for prm in world.parameters():
                prm.mul_(1.5)
        env.sim_world = world
        rec.update(_state_dict_arrays(world, "epi%d_world__" % e))
        np.random.seed(seed)
        env.reset("test")                       # 'train' would make one human: the ORCA robot is not multiagent_training
        rob0, hum0 = G.full_state_rows(env)
        arng = np.random.RandomState(40 + e)
        steps = {key: [] for key in ("act", "look_obs", "look_reward", "look_done", "look_info", "obs", "reward", "done",
                                     "info", "rob", "hum", "time")}
        done = False
        while not done and len(steps["act"]) < 110:
            a = acts[0] if len(steps["act"]) % 7 == 6 else acts[1 + 5 * 4 + int(arng.randint(0, 5))]   # mostly towards +y
            if e % 2:
                a = acts[1 + 5 * 4 + 4]                                # straight to the goal at full speed
            elif arng.rand() < 0.3:
                a = acts[int(arng.randint(0, 81))]
            with torch.no_grad():
                ob, r, d, info = env.onestep_lookahead(ActionXY(*a))
            steps["look_obs"].append([[o.px, o.py, o.vx, o.vy, o.radius] for o in ob])
            steps["look_reward"].append(r); steps["look_done"].append(d); steps["look_info"].append(G.info_code(info))
            with torch.no_grad():
                ob, r, done, info = env.step(ActionXY(*a))
            steps["act"].append(a)
            steps["obs"].append([[o.px, o.py, o.vx, o.vy, o.radius] for o in ob])
            steps["reward"].append(r); steps["done"].append(done); steps["info"].append(G.info_code(info))
            rob, hum = G.full_state_rows(env)
            steps["rob"].append(rob); steps["hum"].append(hum); steps["time"].append(env.global_time)
        rec["epi%d_seed" % e] = np.array(seed); rec["epi%d_rule" % e] = np.array(rule)
        rec["epi%d_rob0" % e], rec["epi%d_hum0" % e] = rob0, hum0
        for key, val in steps.items():
            rec["epi%d_%s" % (e, key)] = np.array(val, np.float64 if key not in ("done", "look_done", "info", "look_info") else np.int32)
        print("  episode %d: %d steps, ends with info %d" % (e, len(steps["act"]), steps["info"][-1]))
    np.savez_compressed(os.path.join(OUT, "g14_model_env.npz"), **rec)
    print("g14_model_env: %d arrays, %d resets" % (len(rec), k))


def g15_explorer():
    """The reference's own Explorer.run_k_episodes (crowd_nav/utils/explorer.py:36-151) on the reference CrowdSim (ORCA
    humans through the rvo2 stand-in).  explorer.py:51 reads `np.NaN`, an alias numpy 2.x removed: this tool process
    (only) puts it back before the call -- the reference file is untouched."""
    import tempfile
    from crowd_nav.utils.explorer import Explorer
    from crowd_nav.utils.memory import ReplayMemory
    if not hasattr(np, "NaN"):
        np.NaN = np.nan
    rec = {}
    dev = torch.device("cpu")
    target = _sarl_policy(51)
    target.kinematics = "holonomic"
    rec.update(_state_dict_arrays(target.model, "target_w__"))

    def memory_arrays(mem, tag):
        if len(mem.memory):
            rec[tag + "_mem_states"] = np.stack([m[0].numpy() for m in mem.memory])
            rec[tag + "_mem_values"] = np.array([float(m[1]) for m in mem.memory], np.float32)
        else:
            rec[tag + "_mem_states"] = np.zeros((0, 5, 13), np.float32)
            rec[tag + "_mem_values"] = np.zeros(0, np.float32)

    # ---- (1) imitation learning: the ORCA demonstrator of train.py:150-160 (safety_space from train.config)
    for tag, phase, k, kw in (("il_val", "val", 12, dict(returnNav=True)),
                              ("il_train", "train", 9, dict(returnRate=False)),
                              ("il_fixed_case", "test", 3, dict(test_case=3, returnNav=True))):
        env, robot, pol = G.make_env("CrowdSim", robot_policy="orca", human_num=5)
        pol.multiagent_training = True
        pol.safety_space = 0.15
        mem = ReplayMemory(100000)
        ex = Explorer(env, robot, dev, mem, 0.9, target_policy=target)
        out = ex.run_k_episodes(k, phase, update_memory=True, imitation_learning=True, **kw)
        rec[tag + "_out"] = np.array(out, np.float64)
        rec[tag + "_counter"] = np.array(env.case_counter[phase])
        memory_arrays(mem, tag)
        print("  %s: out %s, %d memory rows" % (tag, [round(float(x), 4) for x in out], len(mem.memory)))
    # ---- (2) greedy SARL robot (seeded weights), RL value targets from a target network.  Phase 'train' with epsilon 0:
    # the policy only keeps `last_state` in the train phase (multi_human_rl.py:60-61), which update_memory needs
    torch.manual_seed(52)
    env, robot, pol = G.make_env("CrowdSim", robot_policy="sarl", human_num=5)
    rec.update(_state_dict_arrays(pol.model, "sarl_w__"))
    mem = ReplayMemory(100000)
    ex = Explorer(env, robot, dev, mem, 0.9, target_policy=pol)
    ex.update_target_model(target.model)
    pol.set_epsilon(0.0)
    with torch.no_grad():
        out = ex.run_k_episodes(6, "train", update_memory=True, imitation_learning=False, returnNav=True)
    rec["sarl_train_out"] = np.array(out, np.float64)
    memory_arrays(mem, "sarl_train")
    print("  sarl_train: out %s, %d memory rows" % ([round(float(x), 4) for x in out], len(mem.memory)))
    # ---- (3) data collection: robot stays, every step into raw_memory, world-model pairs, one SGAN text file per episode
    for tag, stay in (("collect_stay", True), ("collect_orca", False)):
        env, robot, pol = G.make_env("CrowdSim", robot_policy="orca", human_num=5)
        pol.multiagent_training = True
        ex = Explorer(env, robot, dev, None, 0.9)
        ex.raw_memory, ex.rawob = ReplayMemory(100000), ReplayMemory(100000)
        with tempfile.TemporaryDirectory() as d:
            out = ex.run_k_episodes(4, "val", stay=stay, update_raw_ob=True, cacheFile=d, returnNav=True)
            for i in range(4):
                rec["%s_cache%d" % (tag, i + 1)] = np.array(open(os.path.join(d, "%d.txt" % (i + 1))).read())
        rec[tag + "_out"] = np.array(out, np.float64)
        raw = ex.raw_memory.memory
        rec[tag + "_raw_ob"] = np.array([[[h.px, h.py, h.vx, h.vy, h.radius] for h in r[0]] for r in raw], np.float64)
        rec[tag + "_raw_reward"] = np.array([r[1] for r in raw], np.float64)
        rec[tag + "_raw_done"] = np.array([r[2] for r in raw], np.uint8)
        rec[tag + "_raw_info"] = np.array([G.info_code(r[3]) for r in raw], np.int32)
        rec[tag + "_pairs_cur"] = np.stack([p[0].numpy() for p in ex.rawob.memory])
        rec[tag + "_pairs_next"] = np.stack([p[1].numpy() for p in ex.rawob.memory])
        print("  %s: out %s, %d raw rows, %d pairs" % (tag, [round(float(x), 4) for x in out], len(raw), len(ex.rawob.memory)))
    np.savez_compressed(os.path.join(OUT, "g15_explorer.npz"), **rec)
    print("g15_explorer: %d arrays" % len(rec))


def g16_orca_robot():
    """test.py:64-109 with --policy orca: the reference CrowdSim (visible and invisible robot), ORCA robot and humans through
    the rvo2 stand-in; after the robot has arrived, crowd_sim.py:219-258 runs everybody to the end."""
    rec = {}
    for vis in (False, True):
        env, robot, pol = G.make_env("CrowdSim", robot_policy="orca", human_num=5, robot_visible=vis)
        for case in (0, 3, 6, 11):
            ob = env.reset("test", case)
            done = False
            A, R, I, ST = [], [], [], []
            while not done:
                action = robot.act(ob)
                ob, reward, done, info = env.step(action)
                rob, hum = G.full_state_rows(env)
                A.append([action.vx, action.vy]); R.append(reward); I.append(G.info_code(info))
                ST.append(np.concatenate([rob, hum.ravel()]))
            key = "v%d_c%d_" % (vis, case)
            rec[key + "actions"], rec[key + "rewards"] = np.array(A), np.array(R)
            rec[key + "info"], rec[key + "states"] = np.array(I), np.array(ST)
            rec[key + "time"] = np.array(env.global_time)
            rec[key + "human_times_step"] = np.array(env.human_times, np.float64)
            if I[-1] == G.INFO_CODE["ReachGoal"]:
                ht = env.get_human_times()          # test.py:106: after the robot's last step has put it on the goal
                rob, hum = G.full_state_rows(env)
                rec[key + "human_times"] = np.array(ht, np.float64)
                rec[key + "end_rob"], rec[key + "end_hum"] = rob, hum
                rec[key + "end_time"] = np.array(env.global_time)
                rec[key + "n_states"] = np.array(len(env.states))
            print("  visible %d case %d: %d steps, outcome %d%s" % (vis, case, len(A), I[-1],
                  ", human times %s" % [round(float(x), 2) for x in rec[key + "human_times"]] if key + "human_times" in rec else ""))
    np.savez_compressed(os.path.join(OUT, "g16_orca_robot.npz"), **rec)
    print("g16_orca_robot: %d arrays" % len(rec))


def g17_sarl_unicycle():
    from crowd_sim.envs.utils.state import FullState, ObservableState, JointState
    rng = np.random.RandomState(17)
    rec = {}
    p = _sarl_policy(17)
    p.kinematics = "unicycle"
    rec.update(_state_dict_arrays(p.model, "w__"))
    for N in (5, 10, 2):
        states, vals, acts, atts = [], [], [], []
        for s_ in range(40):
            rpx, rpy = rng.uniform(-3, 3, 2)
            near_goal = s_ % 8 == 7
            gx, gy = (rpx + rng.uniform(-0.2, 0.2), rpy + rng.uniform(-0.2, 0.2)) if near_goal else rng.uniform(-4, 4, 2)
            theta = rng.uniform(-np.pi, 2 * np.pi)
            sp = rng.uniform(0, 1)
            me = FullState(rpx, rpy, sp * np.cos(theta), sp * np.sin(theta), 0.3, gx, gy, 1.0, theta)
            hs = []
            for i in range(N):
                if s_ % 3 == 0 and i < 2:
                    a, d = rng.uniform(0, 2 * np.pi), rng.uniform(0.5, 1.3)
                    hx, hy = rpx + d * np.cos(a), rpy + d * np.sin(a)
                else:
                    hx, hy = rng.uniform(-4, 4, 2)
                hs.append(ObservableState(hx, hy, rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(0.3, 0.5)))
            js = JointState(me, hs)
            with torch.no_grad():
                act = p.predict(js)
            states.append((np.array([me.px, me.py, me.vx, me.vy, me.radius, me.gx, me.gy, me.v_pref, me.theta]),
                           np.array([[h.px, h.py, h.vx, h.vy, h.radius] for h in hs])))
            acts.append([act.v, act.r])
            vals.append(np.array(p.action_values) if not p.reach_destination(js) else np.full(81, np.nan))
            # what env.step stores per step for rendering (crowd_sim.py:411-412): the weights of the LAST forward, i.e.
            # of the last candidate action (sarl.py:56,88-89) -- stale where predict() returned early
            atts.append(np.array(p.get_attention_weights(), np.float32))
        rec["N%d_attention" % N] = np.array(atts)
        rec["N%d_self" % N] = np.array([s_[0] for s_ in states])
        rec["N%d_humans" % N] = np.array([s_[1] for s_ in states])
        rec["N%d_values" % N] = np.array(vals)
        rec["N%d_action" % N] = np.array(acts)
    rec["table"] = np.array([[a.v, a.r] for a in p.action_space])
    np.savez_compressed(os.path.join(OUT, "g17_sarl_unicycle.npz"), **rec)
    print("g17_sarl_unicycle: %d arrays" % len(rec))


def g18_lookahead_in_sim():
    from crowd_sim.envs.utils.action import ActionXY
    from crowd_nav.policy.world_model import MlpWorld
    rec = {}
    torch.manual_seed(18)
    env, robot, pol = G.make_env("CrowdSim", robot_policy="sarl", humans_policy="orca", human_num=5)
    pol.query_env = True
    pol.set_env(env)
    rec.update(_state_dict_arrays(pol.model, "w__"))
    torch.manual_seed(180)
    world = MlpWorld(5).eval()
    with torch.no_grad():
        for prm in world.parameters():
            prm.mul_(1.5)
    rec.update(_state_dict_arrays(world, "world__"))
    env.look_ahead_in_sim = True
    env.sim_world = world
    env.device = torch.device("cpu")
    for case in (1, 4):
        ob = env.reset("test", case)
        done, t = False, 0
        A, V, R, I, LO, LR, LI = [], [], [], [], [], [], []
        while not done and t < 20:
            with torch.no_grad():
                action = robot.act(ob)
                lob, lr, ld, linfo = env.onestep_lookahead(ActionXY(0.3, -0.2))      # the env call on its own
            V.append(np.array(pol.action_values, np.float64))
            LO.append([[o.px, o.py, o.vx, o.vy, o.radius] for o in lob]); LR.append(lr); LI.append(G.info_code(linfo))
            ob, reward, done, info = env.step(action)
            A.append([action.vx, action.vy]); R.append(reward); I.append(G.info_code(info))
            t += 1
        key = "c%d_" % case
        rec[key + "actions"], rec[key + "values"] = np.array(A), np.array(V)
        rec[key + "rewards"], rec[key + "info"] = np.array(R), np.array(I)
        rec[key + "look_obs"], rec[key + "look_reward"], rec[key + "look_info"] = np.array(LO), np.array(LR), np.array(LI)
        print("  look_ahead_in_sim case %d: %d steps, outcome %d" % (case, t, I[-1]))
    np.savez_compressed(os.path.join(OUT, "g18_lookahead_in_sim.npz"), **rec)
    print("g18_lookahead_in_sim: %d arrays" % len(rec))


def g19_sganworld():
    import tempfile
    from crowd_nav.policy.world_model import SGANWorld
    rng = np.random.RandomState(19)
    rec = {}
    for fam, tag in (("sgan-p-models", "p"), ("sgan-models", "np")):
        for case, (N, late) in enumerate(((5, False), (3, True), (10, False))):
            with tempfile.TemporaryDirectory() as d:
                path = os.path.join(d, "generate.txt")
                pos0 = rng.uniform(-4, 4, (N, 2)); vel0 = rng.uniform(-0.8, 0.8, (N, 2))
                lines = []
                for f in range(8):                                     # frames 11..18, 0.25 s apart
                    for p_ in range(N):
                        if late and p_ == N - 1 and f < 3:
                            continue                                   # the last pedestrian enters at the 4th frame
                        x, y = pos0[p_] + vel0[p_] * 0.25 * (f - 7) + rng.normal(0, 0.01, 2)
                        lines.append("%s\t%s\t%s\t%s\n" % (11 + f, p_, x, y))
                open(path, "w").write("".join(lines))
                key = "%s_c%d_" % (tag, case)
                rec[key + "cache0"] = np.array("".join(lines))
                world = SGANWorld(path, torch.device("cpu"), obs_len=8, time_step=0.25,
                                  pretrainPath=os.path.join(REF, "sgan", "models", fam, "zara1_8_model.pt"))
                # the state the env hands over: positions continue the recorded motion
                last = np.array([[float(v) for v in ln.split("\t")[2:4]] for ln in lines if ln.startswith("18\t")])
                pos, vel = last + vel0 * 0.25, vel0.copy()
                ins, outs = [], []
                for step in range(7):
                    torch.manual_seed(1900 + step)
                    in_state = [[pos[i, 0], pos[i, 1], vel[i, 0], vel[i, 1]] for i in range(N)]
                    v = np.asarray(world(in_state), np.float64)
                    ins.append(in_state); outs.append(v)
                    pos, vel = pos + v * 0.25, v
                rec[key + "in"], rec[key + "out"] = np.array(ins), np.array(outs)
                rec[key + "cache_end"] = np.array(open(path).read())
            print("  %s case %d: N %d late %s, |v| up to %.3f" % (tag, case, N, late, np.abs(outs).max()))
    np.savez_compressed(os.path.join(OUT, "g19_sganworld.npz"), **rec)
    print("g19_sganworld: %d arrays" % len(rec))


def g20_trainer_sim():
    """The reference's world-model trainer made reproducible: AttentionWorld has no dropout, so the only random inputs are
    `random.shuffle(memory)` before the 80 / 20 split (trainer_sim.py:55; Python's generator, seeded here) and the two
    DataLoaders' shuffles, which draw from torch's global generator -- recorded by replaying identical DataLoaders over
    row indices from the same generator state (they consume it exactly as the trainer's).
    pytorchtools.py:25 reads `np.Inf`, an alias numpy 2.x removed: restored in this tool process only."""
    import random
    import tempfile
    import torch.utils.data as tud
    from crowd_sim.envs.utils.state import JointState  # noqa: F401
    from crowd_nav.policy.world_model import AttentionWorld
    from crowd_nav.utils.trainer_sim import Trainer_Sim
    from crowd_nav.utils.memory import ReplayMemory
    if not hasattr(np, "Inf"):
        np.Inf = np.inf
    rng = np.random.RandomState(20)
    rec = {}
    N, rows = 5, 240
    cur = rng.uniform(-3, 3, (rows, N, 4)).astype(np.float32)
    nxt = (0.8 * cur[:, :, 2:4] + 0.05 * np.tanh(cur[:, :, 0:2])).astype(np.float32)        # a smooth law to fit
    rec["cur"], rec["next"] = cur, nxt

    class _Rows(tud.Dataset):
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

        def __getitem__(self, i):
            return i

    def replay(n_train, n_val, epochs):
        tl, vl = tud.DataLoader(_Rows(n_train), 1000, shuffle=True), tud.DataLoader(_Rows(n_val), 1000, shuffle=True)
        tp, vp = [], []
        for _ in range(epochs):
            tp.append(np.concatenate([b.numpy() for b in tl]).astype(np.int64))
            vp.append(np.concatenate([b.numpy() for b in vl]).astype(np.int64))
        return np.stack(tp), np.stack(vp)
    # "stop": 30 rows whose validation part (known: the same seeded shuffle of a 30-element list) carries the NEGATED law --
    # the better the training part is fitted, the worse the validation loss: best = first epoch, stop after 7 more
    order = list(range(30))
    random.seed(2000)
    random.shuffle(order)
    nxt_stop = nxt[:30].copy()
    nxt_stop[order[24:]] *= -1.0
    rec["next_stop"] = nxt_stop
    for tag, calls, lr, use in (("short", (5, 3), 1e-3, rows), ("stop", (50,), 1e-3, 30)):
        torch.manual_seed(200)
        model = AttentionWorld()
        if tag == "short":
            rec.update(_state_dict_arrays(model, "w0__"))
        mem = ReplayMemory(10000)
        for i in range(use):
            mem.push((torch.from_numpy(cur[i]), torch.from_numpy((nxt if tag == "short" else nxt_stop)[i])))
        n_train = int(use * 0.8)
        with tempfile.TemporaryDirectory() as d:
            tr = Trainer_Sim(model, mem, torch.device("cpu"), 1000, os.path.join(d, "checkpoint.pt"))
            tr.set_learning_rate(lr)
            random.seed(2000)
            torch.manual_seed(201)
            state = torch.get_rng_state()
            bests = [tr.optimize_epoch(e) for e in calls]            # a second call keeps the best score
        torch.set_rng_state(state)
        for c, e in enumerate(calls):
            tp, vp = replay(n_train, use - n_train, e)
            rec["%s_call%d_train_perms" % (tag, c)], rec["%s_call%d_val_perms" % (tag, c)] = tp, vp
        rec[tag + "_best"] = np.array(bests, np.float64)
        rec[tag + "_mse"] = np.array(model.mse, np.float64)
        rec[tag + "_counter"] = np.array(tr.early_stopping.counter)
        rec[tag + "_stopped"] = np.array(bool(tr.early_stopping.early_stop))
        rec[tag + "_rows"] = np.array(use)
        w1 = _state_dict_arrays(model, tag + "_w1__")
        if tag == "stop":                                   # the small layers are enough to tell the restored weights
            w1 = {k_: v_ for k_, v_ in w1.items() if "mlp3__6" in k_ or "mlp1__0" in k_ or "attention__4" in k_}
        rec.update(w1)
        print("  %s: best %s, mse %.6g, counter %d, stopped %s" % (tag, bests, model.mse, tr.early_stopping.counter,
                                                                 tr.early_stopping.early_stop))
    np.savez_compressed(os.path.join(OUT, "g20_trainer_sim.npz"), **rec)
    print("g20_trainer_sim: %d arrays" % len(rec))


def g21_epsilon():
    from crowd_sim.envs.utils.state import FullState, ObservableState, JointState
    rng = np.random.RandomState(21)
    p = _sarl_policy(17)                         # the weights g17_sarl_unicycle.npz already holds
    g17 = np.load(os.path.join(OUT, "g17_sarl_unicycle.npz"))
    assert all(np.array_equal(v, g17[k]) for k, v in _state_dict_arrays(p.model, "w__").items())
    p.kinematics = "holonomic"
    p.set_phase("train")
    p.set_epsilon(0.5)
    N = 5
    selfs, hums, acts, lasts, explored = [], [], [], [], []
    np.random.seed(2100)
    for s_ in range(60):
        rpx, rpy = rng.uniform(-3, 3, 2)
        gx, gy = (rpx + 0.1, rpy - 0.1) if s_ % 15 == 14 else rng.uniform(-4, 4, 2)
        me = FullState(rpx, rpy, rng.uniform(-1, 1), rng.uniform(-1, 1), 0.3, gx, gy, 1.0, 0.0)
        hs = [ObservableState(*rng.uniform(-4, 4, 2), rng.uniform(-1, 1), rng.uniform(-1, 1), 0.3) for _ in range(N)]
        js = JointState(me, hs)
        p.action_values = None
        with torch.no_grad():
            act = p.predict(js)
        selfs.append([me.px, me.py, me.vx, me.vy, me.radius, me.gx, me.gy, me.v_pref, me.theta])
        hums.append([[h.px, h.py, h.vx, h.vy, h.radius] for h in hs])
        acts.append([act.vx, act.vy])
        lasts.append(p.last_state.numpy().copy())
        # 0: greedy (the look-ahead ran), 1: a random table row, 2: the robot stands on its goal (no draw at all)
        explored.append(2 if p.reach_destination(js) else int(p.action_values is None))
    rec = dict(selfs=np.array(selfs), humans=np.array(hums), actions=np.array(acts), last_states=np.array(lasts),
               explored=np.array(explored), table=np.array([[a.vx, a.vy] for a in p.action_space]))
    np.savez_compressed(os.path.join(OUT, "g21_epsilon.npz"), **rec)
    print("g21_epsilon: %d arrays; greedy / random / at goal: %s" % (len(rec), np.bincount(np.array(explored))))


FAMILIES = {"g21": g21_epsilon, "g20": g20_trainer_sim, "g19": g19_sganworld, "g18": g18_lookahead_in_sim, "g17": g17_sarl_unicycle, "g16": g16_orca_robot, "g15": g15_explorer, "g14": g14_model_env, "g13": g13_world, "g10": g10_trainer, "g11": g11_queryenv, "g12": g12_update_memory, "g5": g5_sarl, "g6": g6_sgan, "g7": g7_episode, "g8": g8_datagen, "g9": g9_realdata}
