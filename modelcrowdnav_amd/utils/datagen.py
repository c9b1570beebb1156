"""Batched DataGen: "explore in mixed reality" rollouts (reference: crowd_nav/utils/datagen.py:379-543).

The reference plays ONE sample at a time: pick a recorded real episode, cut it at a random length, put the
ModelCrowdSim into replay mode (`set_current_state`), let the robot policy act while the humans first replay their
recorded velocities (`env.step(action, new_v=...)`) and are then "imagined" by the world model (`new_v=None`) or
frozen, collect (rotated joint state, reward) pairs, and push value targets into the replay memory when the sample
ends in ReachGoal / Collision.

Here E samples run side by side on a VecModelCrowdSim resident in HBM: the recorded episodes live on the device as
one [episodes, T_max, N, 5] tensor, each step is one policy launch (`predict_batch`), one gather of the recorded
velocities (or one world-model launch) and one `mcn_env_step` launch in given-velocity mode; rewards / dones /
infos stay on the device and the value targets are computed by the same batched routine as VecExplorer
(`rollout.value_targets`).  The host-side random draws (episode choice, cut length) are made with Python's `random`
in the reference's order, so with the same seed and a greedy policy the samples are the reference's samples.

`replace_robot` (the robot takes over one recorded pedestrian's start and goal, datagen.py:262-317) is decided per
sample on the host with the reference's arithmetic and draws; the chosen pedestrian's column is dropped on the device.

`view_human` / `view_distance` (the policy sees only the n closest pedestrians / those within a distance,
datagen.py:346-377) are a stable sort + gather into a shadow state; a ragged count goes to the look-ahead kernel as
`mcn_env_state.hcount`.

Recordings whose crowd grows over time (misc.py `padding_first='none'`, datagen.py:457-466) replay with the pedestrians
that have not entered yet parked far outside the scene and hidden from the policy through `hcount`; when a frame brings
new pedestrians, all pedestrians of that frame are re-placed at their recorded state, as the reference does.

Imagination (`add_sim`) of such a crowd goes through `mcn_sgan_step`'s per-scene counts (VecSGANWorld).

Not carried over this round (raise NotImplementedError): ragged pedestrian counts (`view_distance`, growing crowds)
together with updateMemory (the stored states would be ragged; the reference's own Trainer cannot collate those
either) or with a fixed-width world model (MlpWorld), `render_path`.
"""
import copy
import logging
import math
import random

import numpy as np
import torch

from .. import _hip
from ..rollout import value_targets, average


def _ob_rows(ob):
    """One recorded observation -> [N,5] float64 rows (px, py, vx, vy, radius).  Accepts the reference's
    list[ObservableState] (state.py:27-43) or an array-like."""
    if len(ob) and hasattr(ob[0], "px"):
        return np.array([[h.px, h.py, h.vx, h.vy, h.radius] for h in ob], np.float64)
    return np.asarray(ob, np.float64).reshape(-1, 5)


class _PolicyView(object):
    """What the policy may look at: of the pedestrians present in the env (all N, or the first count[e] when the
    recording's crowd grows over time), `view_distance` keeps those within that distance of the robot (in their
    original order; the closest one if none is) and `view_human` then keeps the n closest, closest first
    (datagen.py:346-377).  The kept pedestrians are gathered to the front of a shadow state and their number goes into
    `hcount`, which the look-ahead kernel honours (mcn_env_state.hcount).  Exposes the slice of the VecCrowdSim surface
    that `predict_batch` / `transform_batch` read."""

    def __init__(self, env, n=-1, distance=-1.0, ragged=False):
        self.env, self.n, self.distance, self.ragged = env, int(n), float(distance), bool(ragged)
        N = env._alloc_N
        masked = self.distance > 0 or self.ragged
        self.width = min(self.n, N) if (self.n > 0 and not masked) else N          # fixed count: no masking needed
        self.num_envs, self.device, self.robot, self._alloc_N = env.num_envs, env.device, env.robot, self.width
        E, dev = env.num_envs, env.device
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float64, device=dev)
        W = self.width
        self.hpos, self.hvel, self.hgoal, self.hrad, self.hvpref = z(E, W, 2), z(E, W, 2), z(E, W, 2), z(E, W), z(E, W)
        for k in ("rpos", "rvel", "rgoal", "rrad", "rvpref", "rtheta", "gtime"):
            setattr(self, k, getattr(env, k))
        self.human_times = z(E, W)
        self.hcount = torch.full((E,), W, dtype=torch.int32, device=dev) if masked else None
        self._st = _hip.EnvState(*[_hip.ptr(t) for t in (self.hpos, self.hvel, self.hgoal, self.hrad, self.hvpref,
                                                         self.rpos, self.rvel, self.rgoal, self.rrad, self.rvpref,
                                                         self.rtheta, self.gtime, self.human_times, self.hcount)])

    def refresh(self, count=None):
        env = self.env
        d = env.hpos - env.rpos.unsqueeze(1)                       # norm([rpx - h.px, rpy - h.py]) per pedestrian
        dist = torch.sqrt(torch.addcmul(d[..., 0] * d[..., 0], d[..., 1], d[..., 1]))
        N = dist.shape[1]
        if self.hcount is not None:
            present = torch.ones_like(dist, dtype=torch.bool) if count is None else \
                torch.arange(N, device=dist.device).view(1, N) < count.view(-1, 1)
            valid = present
            if self.distance > 0:
                valid = present & (dist <= self.distance)
                none = ~valid.any(1)
                masked_d = torch.where(present, dist, torch.full_like(dist, float("inf")))
                closest = torch.zeros_like(valid).scatter_(1, masked_d.argmin(1, keepdim=True), True)
                valid = torch.where(none.unsqueeze(1), closest, valid)
            cnt = valid.sum(1)
            if self.n > 0:                                          # n closest of the visible ones, closest first
                key = torch.where(valid, dist, torch.full_like(dist, float("inf")))
                cnt = cnt.clamp(max=self.n)
            else:                                                   # visible ones in their original order
                key = (~valid).to(dist.dtype)
            idx = torch.argsort(key, dim=1, stable=True)
            self.hcount.copy_(cnt.to(torch.int32))
        else:
            idx = torch.argsort(dist, dim=1, stable=True)
        idx = idx[:, :self.width]
        i2 = idx.unsqueeze(2).expand(-1, -1, 2)
        self.hpos.copy_(torch.gather(env.hpos, 1, i2)); self.hvel.copy_(torch.gather(env.hvel, 1, i2))
        self.hrad.copy_(torch.gather(env.hrad, 1, idx))
        return self


class VecDataGen(object):
    def __init__(self, memory, robot, env, policy):
        """env: a configured VecModelCrowdSim (robot set).  policy: the robot's policy (predict_batch /
        transform_batch surface: SARL and the other MultiHumanRL policies)."""
        self.counter = 0
        self.raw_memory = None
        self.memory = memory
        self.robot = robot
        self.policy = policy
        self.target_model = None
        self.env = env
        self.gamma = policy.gamma
        self._epi = None

    def update_target_model(self, target_model):
        self.target_model = copy.deepcopy(target_model)

    # ------------------------------------------------------------------ recorded episodes
    def get_episode_start_index(self):
        """datagen.py:220-230."""
        indexes, add = [], True
        for i, data in enumerate(self._raw_list()):
            if add:
                indexes.append(i)
                add = False
            if data[2]:
                add = True
        return indexes

    def count(self):
        return len(self.get_episode_start_index())

    def _raw_list(self):
        raw = self.raw_memory
        return raw.memory if hasattr(raw, "memory") else raw

    def load_real_episodes(self, max_human=-1):
        """raw_memory (the reference's list of (ob, reward, done, info[, start_ends]) rows, misc.py:85-89 /
        explorer.py:116-121) -> device tensors: obs [n_epi, T_max, N, 5] float64, length [n_epi]."""
        raw = self._raw_list()
        starts = self.get_episode_start_index()
        epis, ses = [], []
        for s in starts:
            rows = []
            se = None
            for data in raw[s:]:
                ob = _ob_rows(data[0])
                if 0 < max_human < len(ob):
                    ob = ob[:max_human]
                rows.append(ob)
                if len(data) == 5:                       # has start_ends (misc.py:88): the last visited row's copy
                    se = data[4][:max_human] if max_human > 0 else data[4]
                if data[2]:
                    break
            ses.append(None if se is None else np.asarray(se, np.float64).reshape(-1, 4))
            cnt = [r.shape[0] for r in rows]
            if any(b < a for a, b in zip(cnt, cnt[1:])):
                raise NotImplementedError("episode starting at raw_memory[%d]: the crowd shrinks (%s)" % (s, cnt))
            P = max(cnt)
            full = np.zeros((len(rows), P, 5), np.float64)
            for t, r in enumerate(rows):                      # pedestrians enter at the end of the list (misc.py:136-158)
                full[t, :r.shape[0]] = r
            epis.append((full, np.asarray(cnt, np.int64)))
        T = max(e[0].shape[0] for e in epis)
        P = max(e[0].shape[1] for e in epis)
        obs = np.zeros((len(epis), T, P, 5), np.float64)
        count = np.zeros((len(epis), T), np.int64)
        for i, (e, c) in enumerate(epis):
            obs[i, :e.shape[0], :e.shape[1]] = e
            count[i, :len(c)] = c
            count[i, len(c):] = c[-1]
        dev = self.env.device
        ragged = bool((count != P).any())
        self._epi = dict(starts=starts, slot={s: i for i, s in enumerate(starts)},
                         obs=torch.from_numpy(obs).to(dev), length=[e[0].shape[0] for e in epis], max_human=max_human,
                         start_ends=ses, first=[e[0][0, :e[1][0]] for e in epis],
                         count=torch.from_numpy(count).to(dev) if ragged else None)
        return self._epi

    # ------------------------------------------------------------------ sample list (host RNG, reference order)
    def _pick_robot(self, start_end, first_obs, random_robot):
        """datagen.py:262-313: which recorded pedestrian the robot replaces, and the robot's padded start / goal.
        Returns (index, (px, py, gx, gy)) or (None, None) when no pedestrian qualifies.  Same arithmetic, same
        `random` draws (one randrange per attempt when random_robot)."""
        distances = [np.linalg.norm([p[2] - p[0], p[3] - p[1]]) for p in start_end]
        avr_dis = np.average(distances)
        limit = self.env.time_limit * self.robot.v_pref
        possible = [i for i in range(len(distances)) if limit > distances[i] > avr_dis]
        if random_robot is False:                 # longest admissible paths first (note: [-0:] keeps the whole list)
            order = sorted(list(enumerate(distances)), key=lambda x: x[1])[-len(possible):][::-1]
            possible = [c[0] for c in order]
        radius = float(self.robot.radius)
        min_dis = 0
        px = py = gx = gy = set_robot = None
        while min_dis < radius * 4:               # the robot must not start on top of a pedestrian
            if len(possible) == 0:
                return None, None
            set_robot = possible.pop(0) if random_robot is False else possible.pop(random.randrange(len(possible)))
            others = [first_obs[i] for i in range(len(first_obs)) if i != set_robot]
            px, py, gx, gy = [start_end[set_robot][i] for i in range(4)]
            mv = [gx - px, gy - py]
            pad_x = 2 * math.sin(mv[0] / np.linalg.norm(mv))
            pad_y = 2 * math.sin(mv[1] / np.linalg.norm(mv))
            px, py, gx, gy = px - pad_x, py - pad_y, gx + pad_x, gy + pad_y
            init_dis = [np.linalg.norm([px - h[0], py - h[1]]) for h in others]
            if len(init_dis) > 0:
                min_dis = min(init_dis)
        return set_robot, (float(px), float(py), float(gx), float(gy))

    def _draw_samples(self, num_sample, min_end, static_end, add_sim, random_epi, test_case, replace_robot=False,
                      random_robot=True):
        ep = self._epi
        indexes = ep["starts"]
        picks = []
        guard = 0
        while len(picks) < num_sample:
            guard += 1
            if guard > 1000 * max(1, num_sample):
                raise RuntimeError("no recorded episode is longer than min_end=%d" % min_end)
            if random_epi:                                   # datagen.py:236-241
                i = random.choice(indexes)
            else:
                i = indexes[self.counter % len(indexes)]
                self.counter += 1
            if test_case is not None:
                i = test_case
                if i not in ep["slot"]:
                    raise NotImplementedError("test_case must be the raw_memory index of an episode start")
            L = ep["length"][ep["slot"][i]]
            robot = (None, None)
            if replace_robot:                                # :262-317, before the length test as in get_real_state
                se = ep["start_ends"][ep["slot"][i]]
                if se is None:
                    raise ValueError("replace_robot needs raw_memory rows with start_ends (misc.py:85-89)")
                robot = self._pick_robot(se.tolist(), ep["first"][ep["slot"][i]].tolist(), random_robot)
                if robot[0] is None:                         # `raw_states == []` -> continue (:409-410)
                    continue
            if L <= min_end:                                 # :412-413
                continue
            length = L
            if add_sim:                                      # :415-418
                length = random.randrange(min_end, L)
                if static_end > 0:
                    length = static_end
            picks.append((ep["slot"][i], min(length, L)) + (robot if replace_robot else ()))
        return picks

    # ------------------------------------------------------------------ the batched loop
    def gen_data_from_explore_in_mix(self, num_sample, phase="train", min_end=1, static_end=-1, max_human=-1,
                                     imitation_learning=False, add_sim=True, stay=False, random_epi=True,
                                     random_robot=True, render_path=None, view_distance=-1, view_human=-1,
                                     returnRate=False, updateMemory=True, replace_robot=False, sgan_genfile=None,
                                     test_case=None, returnNav=False):
        """Same arguments and return value as datagen.py:379-518.  `sgan_genfile` (the text file that seeds the
        SGAN world model's history in the reference, :421-430) is honoured by seeding the HBM history ring of a
        VecSGANWorld with the last `min_end` real frames of every sample; its value is otherwise unused."""
        if render_path is not None:
            raise NotImplementedError("render_path is not carried over")
        if view_distance > 0 and updateMemory and not stay:
            raise NotImplementedError("view_distance with updateMemory would store ragged states")
        env, pol = self.env, self.policy
        if self._epi is None or self._epi["max_human"] != max_human:
            self.load_real_episodes(max_human)
        ep = self._epi
        E, dev = env.num_envs, env.device
        N = ep["obs"].shape[2]
        pol.set_phase(phase)
        picks = self._draw_samples(num_sample, min_end, static_end, add_sim, random_epi, test_case, replace_robot,
                                   random_robot)
        ragged = ep["count"] is not None                  # the recorded crowd grows over time (datagen.py:457-466)
        sim_counts = False
        if ragged and add_sim:
            import inspect
            sim_counts = env.sim_world is not None and "hcount" in inspect.signature(env.sim_world.__call__).parameters
            if not sim_counts:
                raise NotImplementedError("imagining a crowd whose size changes needs a world model that takes "
                                          "per-scene counts (VecSGANWorld); MlpWorld has a fixed input width")
        if ragged and updateMemory and not stay:
            raise NotImplementedError("a growing crowd with updateMemory would store ragged states")
        if replace_robot:
            N -= 1                                        # the replaced pedestrian's column is dropped
        horizon = int(round(env.time_limit / env.time_step)) + 2
        v_pref, dt = float(self.robot.v_pref), float(env.time_step)
        gbar = pow(self.gamma, dt * v_pref)
        human_radius = float(env._human_radius)
        rec = dict(ret=[], info=[], time=[], steps=[])
        too_close, min_dist = 0, []
        sim = env.sim_world
        for w0 in range(0, num_sample, E):
            wave = picks[w0:w0 + E]
            n_live = len(wave)
            pad = wave + [wave[-1]] * (E - n_live)                       # idle envs repeat the last sample
            slot = torch.tensor([p[0] for p in pad], device=dev)
            length = torch.tensor([p[1] for p in pad], device=dev)
            obs = ep["obs"][slot]                                         # [E,T,N,5]
            T_rec = obs.shape[1]
            cnt_all = ep["count"][slot] if ragged else None               # [E,T] pedestrians present per frame
            rpos = rgoal = None
            if replace_robot:
                # drop the replaced pedestrian's column (:315-317); the robot starts / ends at the padded track ends
                drop = torch.tensor([p[2] for p in pad], device=dev).view(E, 1)
                cols = torch.arange(N, device=dev).view(1, N).expand(E, N)
                cols = cols + (cols >= drop).long()
                obs = torch.gather(obs, 2, cols.view(E, 1, N, 1).expand(E, T_rec, N, 5))
                info_t = torch.tensor([p[3] for p in pad], dtype=torch.float64, device=dev)
                rpos, rgoal = info_t[:, 0:2].contiguous(), info_t[:, 2:4].contiguous()
                if ragged:                                # `ob[:h] + ob[h+1:] if len(ob) > h else ob` (:317)
                    cnt_all = cnt_all - (cnt_all > drop).long()
            pos0, vel0 = obs[:, 0, :, 0:2].contiguous(), obs[:, 0, :, 2:4].contiguous()
            cur_cnt = None
            if ragged:
                # pedestrians that have not entered yet wait far outside the scene; the policy never sees them
                # (hcount) and they cannot come near the robot
                cur_cnt = cnt_all[:, 0].clone()
                slots = torch.arange(N, device=dev).view(1, N)
                parked = torch.stack([1000.0 + 10.0 * slots.expand(E, N).double(),
                                      torch.full((E, N), 1000.0, dtype=torch.float64, device=dev)], 2)
                absent0 = slots >= cur_cnt.view(E, 1)
                pos0 = torch.where(absent0.unsqueeze(2), parked, pos0)
                vel0 = torch.where(absent0.unsqueeze(2), torch.zeros_like(vel0), vel0)
            env.set_current_state(pos0, vel0, torch.full((E, N), human_radius, dtype=torch.float64, device=dev),
                                  rpos, rgoal)
            hist0 = None
            if add_sim and sgan_genfile is not None and hasattr(sim, "reset_history"):
                # the reference writes raw_states[-min_end:] of the cut episode (:423-430); frames before the
                # episode start repeat its first frame (data_loader pads a short track with its first position)
                k = torch.arange(-(sim.hist.shape[1]), 0, device=dev).view(1, -1) + length.view(-1, 1)
                k = torch.maximum(k, (length - min_end).clamp(min=0).view(-1, 1))
                k = k.view(E, -1, 1).expand(E, k.shape[1], N)
                if ragged:
                    # a pedestrian that enters inside the window stands at its first recorded position before that
                    # (SGANWorld.data_loader pads a late track with its first sample, world_model.py:166-178)
                    enters = (cnt_all.unsqueeze(2) <= slots.view(1, 1, N)).sum(1)            # [E,N] first frame present
                    k = torch.maximum(k, enters.clamp(max=T_rec - 1).unsqueeze(1))
                hist0 = torch.gather(obs[..., 0:2], 1, k.unsqueeze(3).expand(E, k.shape[1], N, 2))
                sim.reset_history(hist0)
            states, rewards, dones, infos = [], [], [], []
            alive = torch.ones(E, dtype=torch.bool, device=dev)
            view = _PolicyView(env, view_human, view_distance, ragged) \
                if ((view_human > 0 or view_distance > 0 or ragged) and not stay) else None
            for i in range(horizon):
                seen = view.refresh(cur_cnt) if view is not None else env   # what the policy sees and the memory stores
                states.append(pol.transform_batch(seen))
                if stay:
                    act = torch.zeros(E, 2, dtype=torch.float64, device=dev)
                else:
                    # epsilon-greedy in phase 'train' happens inside predict_batch (multi_human_rl.py:27-29)
                    act, _ = pol.predict_batch(seen, hcount=getattr(seen, "hcount", None))
                replay = (i + 1) < length                                        # :452 per env
                nxt = obs[:, min(i + 1, T_rec - 1), :, 2:4]
                if ragged:                                 # `[...][:self.env.human_num]`: only who is there moves (:453)
                    nxt = torch.where((slots < cur_cnt.view(E, 1)).unsqueeze(2), nxt, torch.zeros_like(nxt))
                if add_sim:
                    if bool((~replay).any()):
                        imagined = sim(env.hpos, hcount=cur_cnt.to(torch.int32)) if sim_counts else sim(env.hpos)
                        if ragged:                         # slots nobody stands in stay parked
                            imagined = torch.where((slots < cur_cnt.view(E, 1)).unsqueeze(2), imagined,
                                                   torch.zeros_like(imagined))
                        new_v = torch.where(replay.view(E, 1, 1), nxt, imagined)
                        if hist0 is not None and bool(replay.any()):
                            self._restore_history(sim, hist0, replay)
                    else:
                        new_v = nxt
                else:
                    new_v = torch.where(replay.view(E, 1, 1), nxt, torch.zeros_like(nxt))   # humans stop moving
                env.step(act, new_v=new_v.contiguous())
                if ragged:
                    # "add more human when needed" (:457-466): when the next recorded frame has more pedestrians,
                    # every pedestrian of that frame is (re)placed at its recorded position and velocity
                    nxt_cnt = torch.where(replay, cnt_all[:, min(i + 1, T_rec - 1)], cur_cnt)
                    grow = nxt_cnt > cur_cnt
                    if bool(grow.any()):
                        put = (grow.view(E, 1) & (slots < nxt_cnt.view(E, 1))).unsqueeze(2)
                        frame = obs[:, min(i + 1, T_rec - 1)]
                        env.hpos.copy_(torch.where(put, frame[..., 0:2], env.hpos))
                        env.hvel.copy_(torch.where(put, frame[..., 2:4], env.hvel))
                    cur_cnt = nxt_cnt
                rewards.append(env.reward.clone()); dones.append(env.done.bool() & alive)
                infos.append(env.info.clone())
                alive = alive & ~env.done.bool()
                if i % 8 == 7 and not bool(alive[:n_live].any()):
                    break
            if bool(alive[:n_live].any()):
                raise RuntimeError("a sample did not end within %d steps" % horizon)
            R, D, I = torch.stack(rewards), torch.stack(dones), torch.stack(infos)       # [T,E]
            Tn = R.shape[0]
            end = D.float().argmax(0)                                                  # step index of the done
            tt = torch.arange(Tn, device=dev).view(-1, 1)
            in_ep = tt <= end.view(1, -1)
            disc = torch.tensor([pow(self.gamma, t * dt * v_pref) for t in range(Tn)], dtype=torch.float64, device=dev)
            ret = (R * disc.view(-1, 1) * in_ep).sum(0)
            fin = I.gather(0, end.view(1, -1)).squeeze(0)
            danger = (I == _hip.INFO_DANGER) & in_ep
            live = torch.zeros(E, dtype=torch.bool, device=dev); live[:n_live] = True
            too_close += int((danger & live.view(1, -1)).sum().item())
            # dmin of the Danger steps: kept on the device by the step record
            rec["ret"] += ret[:n_live].tolist(); rec["info"] += fin[:n_live].tolist()
            rec["steps"] += (end[:n_live] + 1).tolist()
            if updateMemory:
                if self.memory is None or self.gamma is None:
                    raise ValueError("Memory or gamma value is not set!")
                s, v = value_targets(torch.stack(states), R, D & live.view(1, -1), I, imitation_learning, gbar,
                                     self.target_model, dev)
                self.memory.push_batch(s, v)
            self._last_wave = dict(states=torch.stack(states), rewards=R, dones=D, infos=I, live=n_live)
        k = num_sample
        infos_l = [int(c) for c in rec["info"]]
        reach_goal = sum(1 for c in infos_l if c == _hip.INFO_REACHGOAL)
        collision = sum(1 for c in infos_l if c == _hip.INFO_COLLISION)
        times = [n * dt for n in rec["steps"]]
        success_times = [t for t, c in zip(times, infos_l) if c == _hip.INFO_REACHGOAL]
        avg_nav_time = sum(success_times) / len(success_times) if success_times else env.time_limit
        avg_ret = average(rec["ret"])
        logging.info("Exp in mix has success rate: %.2f, collision rate: %.2f, nav time: %.2f, total reward: %.4f",
                     reach_goal / k, collision / k, avg_nav_time, avg_ret)
        self.last_records = dict(returns=rec["ret"], infos=infos_l, times=times, danger_steps=too_close, samples=picks)
        if returnRate and returnNav:
            return avg_ret, reach_goal / k, collision / k, (k - reach_goal - collision) / k, avg_nav_time
        if returnRate:
            return avg_ret, reach_goal / k, collision / k, (k - reach_goal - collision) / k
        return avg_ret, reach_goal, collision, (k - reach_goal - collision)

    @staticmethod
    def _restore_history(sim, hist0, keep):
        """Envs that are still replaying must not see the frame the world-model call just pushed: the reference
        only touches the history file when it imagines (world_model.py:234-249).  Rewrites their rows of the ring
        (oldest-first order starts at sim.oldest)."""
        H = sim.hist.shape[1]
        order = (torch.arange(H, device=hist0.device) + sim.oldest) % H
        from ..policy.world_model import round4
        rows = keep.nonzero().squeeze(1)
        sim.hist[rows.view(-1, 1), order.view(1, -1)] = round4(hist0[rows].to(sim.hist.dtype))


class DataGen(VecDataGen):
    """The reference's name (datagen.py:20; train_model_based_sgan.py:223 `DataGen(memory, robot, env_sim, policy)`).
    `env` may be the gym-style E = 1 ModelCrowdSim view, whose batched env is then used, or a VecModelCrowdSim."""

    def __init__(self, memory, robot, env, policy):
        vec = env.__dict__.get("_vec") if hasattr(env, "__dict__") else None
        super().__init__(memory, robot, vec if vec is not None else env, policy)
