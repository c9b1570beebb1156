"""Batched Explorer: run k evaluation episodes over E environments resident in HBM
(reference: crowd_nav/utils/explorer.py:36-151, run_k_episodes).

The reference loop is `for i in range(k): ob = env.reset(); while not done: a = robot.act(ob);
ob, r, done, info = env.step(a)`.  Here episode i runs in env (i mod E): every env starts with case
`first + e`, and when it finishes an episode env_step.hip resets it in place from an HBM-resident pool
to case `+E`, so the k episodes use exactly the cases the sequential loop would have used.  Per step
there is one policy launch and one env launch, no host synchronisation; discounted returns
(explorer.py:124), outcomes, navigation times and the "too close" statistics (:88-90) are accumulated
in-kernel.  With several ranks, envs are sharded by global env id and the records are combined by ONE
all_gather (dist.gather_records).
"""
import logging
import os

import numpy as np
import torch

from . import _hip
from . import dist as mdist
from .envs import scenarios as S


def average(xs):
    return sum(xs) / len(xs) if xs else 0


def value_targets(states, rewards, dones, infos, imitation_learning, gamma_bar, target_model=None, device=None,
                  return_index=False):
    """explorer.py:153-186 (= datagen.py:520-543) for a whole batched rollout.  states [T,E,N,13] f32, rewards
    [T,E] f64, dones [T,E] bool, infos [T,E] u8, gamma_bar = gamma ** (time_step * v_pref).  Returns
    (states [M,N,13], values [M]) of the steps that belong to episodes ending in ReachGoal or Collision (only
    those feed the memory, explorer.py:107-110 / datagen.py:482-484)."""
    T, E = rewards.shape
    gbar = gamma_bar
    keep = torch.zeros(T, E, dtype=torch.bool, device=rewards.device)
    values = torch.zeros(T, E, dtype=torch.float64, device=rewards.device)
    good_end = (infos == _hip.INFO_REACHGOAL) | (infos == _hip.INFO_COLLISION)
    if not imitation_learning:
        with torch.no_grad():
            nxt = target_model(states[1:].reshape(-1, states.shape[2], states.shape[3]).to(device or states.device))
        nxt = torch.cat([nxt.view(T - 1, E).double(), torch.zeros(1, E, dtype=torch.float64, device=nxt.device)], 0)
    ep_ok = torch.zeros(E, dtype=torch.bool, device=rewards.device)      # episode containing step t ends well
    run = torch.zeros(E, dtype=torch.float64, device=rewards.device)     # discounted tail sum (IL)
    for t in range(T - 1, -1, -1):
        d = dones[t]
        ep_ok = torch.where(d, good_end[t], ep_ok)          # a done at t starts (going backwards) a new episode
        if imitation_learning:
            run = torch.where(d, rewards[t], rewards[t] + gbar * run)
            values[t] = run
        else:
            values[t] = torch.where(d, rewards[t], rewards[t] + gbar * nxt[t])
        keep[t] = ep_ok
    # steps after the last done of an env belong to an unfinished episode: ep_ok is False there by construction
    idx = keep.t().reshape(-1).nonzero().squeeze(1)           # (env, time) order
    flat_states = states.permute(1, 0, 2, 3).reshape(T * E, states.shape[2], states.shape[3])
    flat_values = values.t().reshape(-1)
    if return_index:           # positions of the kept rows in the (env, time)-flattened trace: env * T + t
        return flat_states[idx], flat_values[idx].float(), idx
    return flat_states[idx], flat_values[idx].float()


class VecExplorer(object):
    def __init__(self, env, robot, device=None, gamma=0.9, policy=None, memory=None, target_policy=None):
        self.env, self.robot, self.gamma = env, robot, gamma
        self.device = device or env.device
        self.policy = policy if policy is not None else robot.policy
        self.memory = memory
        self.target_policy = target_policy       # supplies transform() in imitation learning (explorer.py:163)
        self.target_model = None
        self.raw_memory = None                   # list / ReplayMemory-like: rows (ob [N,5], reward, done, info code)
        self.rawob = None                        # list / ReplayMemory-like: (humans [N,4], next velocities [N,2])
        # True: raw_memory rows carry the reference's value types, (list[ObservableState], reward, done, Info object)
        # as explorer.py:80-81 pushes them (the drop-in Explorer sets it); False: arrays and integer codes
        self.raw_rows_as_objects = False

    def update_target_model(self, target_model):
        import copy
        self.target_model = copy.deepcopy(target_model)

    def _value_targets(self, states, rewards, dones, infos, imitation_learning):
        gbar = pow(self.gamma, self.env.time_step * float(self.robot.v_pref))
        return value_targets(states, rewards, dones, infos, imitation_learning, gbar, self.target_model, self.device,
                             return_index=True)

    def _emit_collected(self, cur, ob, rew, done, info, dmin, k, rounds, E_total, update_raw_ob, cache_dir):
        """Split the recorded [T,E,...] traces at the dones and hand the k episodes out in the reference's order
        (episode g = round * E + env): raw_memory rows, world-model pairs, SGAN cache files."""
        T, E = done.shape
        bounds = []
        for e in range(E):
            ends = np.nonzero(done[:, e])[0][:rounds]
            starts = np.concatenate([[0], ends[:-1] + 1])
            bounds.append(list(zip(starts.tolist(), ends.tolist())))
        push = lambda store, item: (store.push if hasattr(store, "push") else store.append)(item)
        fcount = 0
        for g in range(k):
            e, r = g % E_total, g // E_total
            t0, t1 = bounds[e][r]
            frames = []
            for t in range(t0, t1 + 1):
                if self.raw_memory is not None and self.raw_rows_as_objects:
                    from .envs.utils import info as I
                    from .envs.utils.state import ObservableState
                    push(self.raw_memory, ([ObservableState(*[float(x) for x in row]) for row in ob[t, e]],
                                           float(rew[t, e]), bool(done[t, e]), I.from_code(info[t, e], float(dmin[t, e]))))
                elif self.raw_memory is not None:
                    push(self.raw_memory, (ob[t, e].copy(), float(rew[t, e]), bool(done[t, e]), int(info[t, e])))
                if update_raw_ob and (np.abs(ob[t, e, :, 2:4]) > 1e-3).any():        # someone_is_moving
                    push(self.rawob, (torch.from_numpy(cur[t, e]).float(), torch.from_numpy(ob[t, e, :, 2:4].copy()).float()))
                fid = 10 * (t - t0 + 1)
                frames += [[fid, p, ob[t, e, p, 0], ob[t, e, p, 1]] for p in range(ob.shape[2])]
            if cache_dir is not None:
                fcount += 1
                with open(os.path.join(cache_dir, str(fcount) + ".txt"), "w") as fh:
                    for fr in frames:
                        fh.write("%s\t%s\t%s\t%s\n" % (fr[0], fr[1], fr[2], fr[3]))

    def _actions(self, step_actions):
        if step_actions is not None:
            return step_actions
        a, _ = self.policy.predict_batch(self.env)
        return a

    def run_k_episodes(self, k, phase, update_memory=False, imitation_learning=False, episode=None,
                       print_failure=False, returnRate=True, returnNav=False, action_fn=None, max_steps=None,
                       total_envs=None, action_seq=None, device_scenarios=None, stay=False, update_raw_ob=False,
                       cacheFile=None, test_case=None):
        """Returns what Explorer.run_k_episodes returns (explorer.py:146-151):
        (avg cumulative reward, success rate, collision rate, timeout rate[, avg nav time])
        or counts instead of rates when returnRate is False.  `action_fn(env, t) -> [E,2]` overrides the
        policy (e.g. a random-action baseline).  `action_seq` ([T,E,2] device tensor) is a robot whose actions do
        not depend on the observation: its steps go to the device 128 at a time through mcn_env_rollout (one launch,
        state in registers) instead of one launch per step; not combinable with update_memory.
        `device_scenarios=seed` builds the k scenarios on the device (mcn_scenario_pool: the reference's placement
        rules, counter-based random stream -- for training rollouts that need many distinct cases, not for parity
        runs) instead of generating them on the host with numpy's MT19937.
        Data collection (explorer.py:60-85,112-121), single process only: `stay` keeps the robot still; with
        `self.raw_memory` set every step pushes `(ob, reward, done, info)` (ob = [N,5] array of the humans after the
        step, info = code) in episode order; `update_raw_ob` pushes world-model pairs into `self.rawob`; `cacheFile`
        (a directory) gets one SGAN text file per episode.
        `test_case` plays that one case k times (explorer.py:54 hands it to every reset; the counter ends one past it)."""
        env = self.env
        rank, ws = mdist.world()
        E_local = env.num_envs
        E_total = total_envs if total_envs is not None else E_local * ws
        lo, hi = mdist.shard(E_total, rank, ws)                # shards may differ by one env (E_total % ws != 0)
        if hi - lo != E_local:
            raise ValueError("rank %d of %d holds %d envs; its shard of %d is [%d, %d)" % (rank, ws, E_local, E_total, lo, hi))
        if hasattr(self.policy, "set_phase"):
            self.policy.set_phase(phase)
        n, rule = env._phase_rule(phase)
        first = env.case_counter[phase]
        size = env.case_size[phase]
        rounds = -(-k // E_total)                               # episodes per env (ceil)
        cases = [(first + i) % size for i in range(rounds * E_total)]
        if test_case is not None:
            if device_scenarios is not None:
                raise NotImplementedError("test_case with device scenarios")
            cases = [test_case] * (rounds * E_total)
        uniq = sorted(set(cases))
        if device_scenarios is None:
            pool = S.scenario_pool(env.spec(), phase, uniq, n, rule)
            slot_of = {c: j for j, c in enumerate(uniq)}
        else:
            if rule not in ("circle_crossing", "square_crossing"):
                raise NotImplementedError("device scenarios: circle_crossing / square_crossing only")
            # case ids keyed like the reference's seeds: offset(phase) + case (crowd_sim.py:270-272)
            offset = {"train": env.case_capacity["val"] + env.case_capacity["test"], "val": 0,
                      "test": env.case_capacity["val"]}[phase]
            pool = env.device_pool(device_scenarios, offset + uniq[0], uniq[-1] - uniq[0] + 1, n, rule)
            slot_of = {c: c - uniq[0] for c in uniq}
        # env e of this rank plays global episodes (lo + e) + r * E_total, r = 0..rounds-1
        mine = np.array([[slot_of[cases[(lo + e) + r * E_total]] for r in range(rounds)] for e in range(E_local)])
        if device_scenarios is None:
            env.load_scenarios(pool[mine[:, 0]])
        else:
            env.load_device_scenarios(pool, mine[:, 0])
        stride = 0
        if rounds > 1:
            # pool slots advance by a constant stride when cases are consecutive (the usual situation)
            n_pool = len(uniq) if device_scenarios is None else uniq[-1] - uniq[0] + 1
            d = (mine[:, 1] - mine[:, 0]) % n_pool
            if not np.all(d == d[0]) or any(np.any((mine[:, r + 1] - mine[:, r]) % n_pool != d[0])
                                            for r in range(rounds - 1)):
                # the case list wraps unevenly (e.g. the counter starts near case_size): lay the pool out in episode
                # order instead, one row per global episode -- then every env advances by exactly E_total rows
                if device_scenarios is not None:
                    raise NotImplementedError("device scenarios with a case list that wraps unevenly")
                pool = pool[[slot_of[c] for c in cases]]
                mine = np.array([[(lo + e) + r * E_total for r in range(rounds)] for e in range(E_local)])
                d = np.array([E_total % (rounds * E_total)])
            stride = int(d[0])
        # at least two finished-episode slots: with one the kernel keeps the LATEST episode of an env (mcn.h), and an env
        # that finishes early keeps replaying its case until the slowest env is done -- with a stochastic robot
        # (epsilon-greedy, random action_fn) the record would then describe the last repeat instead of the first run
        # the "too close" counters cover exactly the k episodes: env g (global) plays `rounds` of them if
        # (rounds - 1) * E_total + g < k, else one fewer
        n_full = min(max(k - (rounds - 1) * E_total - lo, 0), E_local)
        bufs = env.attach_rollout(self.gamma, pool=pool, case_stride=stride,
                                  first_cases=(mine[:, 1] if rounds > 1 else mine[:, 0]), fin_slots=max(rounds, 2),
                                  danger_episodes=rounds, danger_short_from=(0 if n_full == E_local else n_full + 1))
        horizon = int(round(env.time_limit / env.time_step)) + 2
        limit = max_steps if max_steps is not None else rounds * horizon
        if update_memory and (self.memory is None or self.gamma is None):
            raise ValueError("Memory or gamma value is not set!")
        transformer = (self.target_policy if imitation_learning else self.policy) if update_memory else None
        rec_s, rec_r, rec_d, rec_i = [], [], [], []
        collect = self.raw_memory is not None or update_raw_ob or cacheFile is not None
        if collect:
            if ws > 1 or action_seq is not None:
                raise NotImplementedError("data collection runs in one process, one launch per step")
            keep_export, env.export_human_actions = env.export_human_actions, True
            col_cur, col_ob, col_r, col_d, col_i, col_m = [], [], [], [], [], []
        t = 0
        if action_seq is not None:
            if update_memory:
                raise ValueError("action_seq rollouts record no per-step states; use action_fn with update_memory")
            limit = min(limit, int(action_seq.shape[0]))
            while t < limit:
                n = min(128, limit - t)                      # a launch's time is set by its slowest env group
                env.rollout(action_seq[t:t + n])
                t += n
                if int(bufs["fin_count"].min().item()) >= rounds:
                    break
            limit = t                                            # skip the per-step loop below
        while t < limit:
            if update_memory:
                rec_s.append(transformer.transform_batch(env))           # the state the action is chosen in
            if stay:
                a = torch.zeros(E_local, 2, dtype=torch.float64, device=env.device)
            else:
                a = action_fn(env, t) if action_fn is not None else self._actions(None)
            if collect:
                prev_pos = env.hpos.clone()
                col_cur.append(torch.cat([prev_pos, env.hvel], 2))
            env.step(a)
            if collect:
                # the observation the reference's step() returns, also for envs that finished and were restarted
                # in-kernel: humans moved by the velocity they chose (same two roundings as the kernel's integrate)
                pos = prev_pos + env.human_act * env.time_step
                col_ob.append(torch.cat([pos, env.human_act, env.hrad.unsqueeze(2)], 2))
                col_r.append(env.reward.clone()); col_d.append(env.done.bool()); col_i.append(env.info.clone())
                col_m.append(env.dmin.clone())
            if update_memory:
                rec_r.append(env.reward.clone()); rec_d.append(env.done.bool()); rec_i.append(env.info.clone())
            t += 1
            if t % 32 == 0 and int(bufs["fin_count"].min().item()) >= rounds:
                break
        if int(bufs["fin_count"].min().item()) < rounds:
            raise RuntimeError("rollout did not finish %d episodes per env within %d steps" % (rounds, limit))
        if update_memory:
            # only the first `rounds` episodes of each env are the k requested ones: cut each env's trace there
            dones = torch.stack(rec_d)
            order = torch.cumsum(dones.long(), 0) - dones.long()          # episode number each step belongs to
            valid = order < rounds
            gidx = (order * E_total + lo + torch.arange(E_local, device=dones.device).unsqueeze(0))
            valid &= gidx < k
            s, v, idx = self._value_targets(torch.stack(rec_s), torch.stack(rec_r), dones & valid, torch.stack(rec_i),
                                            imitation_learning)
            # the reference pushes episode by episode (explorer.py:107-110): rows go to the memory in global episode
            # order (episode g = round * E_total + env), time order inside an episode -- a stable sort of the
            # (env, time)-ordered rows by their episode number
            ep_of_row = gidx.t().reshape(-1)[idx]
            perm = torch.argsort(ep_of_row, stable=True)
            s, v = s[perm], v[perm]
            if hasattr(self.memory, "push_batch"):
                self.memory.push_batch(s, v)
            else:                                                         # any object with the reference's push()
                for row_s, row_v in zip(s, v):
                    self.memory.push((row_s, row_v.reshape(1).to(self.device)))
        if collect:
            env.export_human_actions = keep_export
            self._emit_collected(torch.stack(col_cur).cpu().numpy(), torch.stack(col_ob).cpu().numpy(),
                                 torch.stack(col_r).cpu().numpy(), torch.stack(col_d).cpu().numpy(),
                                 torch.stack(col_i).cpu().numpy(), torch.stack(col_m).cpu().numpy(), k, rounds, E_total,
                                 update_raw_ob, cacheFile)
        env.case_counter[phase] = (first + k) % size if test_case is None else (test_case + 1) % size
        # records in global episode order: episode g = r * E_total + global_env
        # (the envs' "too close" counters ride in the same collective, in the rows of their first episode)
        dng = torch.zeros(2, E_local, rounds, dtype=torch.float64, device=bufs["fin_return"].device)
        dng[0, :, 0], dng[1, :, 0] = bufs["danger_count"].double(), bufs["danger_dist_sum"]
        rec = mdist.gather_records(bufs["fin_return"][:rounds].t().contiguous(), bufs["fin_info"][:rounds].t().contiguous(),
                                   bufs["fin_time"][:rounds].t().contiguous(), extras=(dng[0], dng[1]),
                                   equal_shards=(E_total % ws == 0))
        ret = rec["return"].view(-1, rounds).cpu().numpy()        # [E_total, rounds]
        inf = rec["info"].view(-1, rounds).cpu().numpy()
        tim = rec["time"].view(-1, rounds).cpu().numpy()
        order = [(g % E_total, g // E_total) for g in range(k)]
        returns = [float(ret[e, r]) for e, r in order]
        infos = [int(inf[e, r]) for e, r in order]
        times = [float(tim[e, r]) for e, r in order]
        success = sum(1 for c in infos if c == _hip.INFO_REACHGOAL)
        collision = sum(1 for c in infos if c == _hip.INFO_COLLISION)
        timeout = sum(1 for c in infos if c == _hip.INFO_TIMEOUT)
        assert success + collision + timeout == k
        success_times = [tm for tm, c in zip(times, infos) if c == _hip.INFO_REACHGOAL]
        avg_nav_time = sum(success_times) / len(success_times) if success_times else env.time_limit
        extra = "" if episode is None else "in episode {} ".format(episode)
        if not stay:                                                       # explorer.py:132
            logging.info("%-5s %shas success rate: %.2f, collision rate: %.2f, nav time: %.2f, total reward: %.4f",
                         phase.upper(), extra, success / k, collision / k, avg_nav_time, average(returns))
        if print_failure:
            logging.info("Collision cases: %s", " ".join(str(i) for i, c in enumerate(infos) if c == _hip.INFO_COLLISION))
            logging.info("Timeout cases: %s", " ".join(str(i) for i, c in enumerate(infos) if c == _hip.INFO_TIMEOUT))
        # "too close" statistics of the k episodes (explorer.py:88-90,138-141)
        too_close, dist_sum = int(round(rec["extras"][0].sum().item())), float(rec["extras"][1].sum().item())
        if phase in ("val", "test"):
            num_step = sum(times) / env.time_step
            logging.info("Frequency of being in danger: %.2f and average min separate distance in danger: %.2f",
                         too_close / num_step, dist_sum / too_close if too_close else 0.0)
        self.last_records = dict(returns=returns, infos=infos, times=times, danger_steps=too_close,
                                 danger_dist_sum=dist_sum)
        env.detach_rollout()
        if returnRate and returnNav:
            return average(returns), success / k, collision / k, (k - success - collision) / k, avg_nav_time
        if returnRate:
            return average(returns), success / k, collision / k, (k - success - collision) / k
        return average(returns), success, collision, (k - success - collision)
