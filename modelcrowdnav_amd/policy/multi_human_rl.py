"""One-step look-ahead over the action table for pairwise-state value networks
(reference: crowd_nav/policy/multi_human_rl.py:7-104).

`predict(JointState)` keeps the reference's contract -- including the draw from numpy's global
stream before the epsilon test, the reach-destination short cut, `action_values`, `last_state`
and the ValueError on an all-NaN network -- but the 81-iteration Python loop (propagate,
compute_reward, 5 tiny tensors, rotate, forward, .item()) is ONE mcn_sarl_lookahead launch.
`predict_batch(env)` is the same launch over all E environments of a VecCrowdSim, reading the
env's HBM-resident state in place.
"""
import ctypes as C

import numpy as np
import torch

from .. import _hip
from .cadrl import CADRL
from ..envs.utils.action import ActionRot, ActionXY


class MultiHumanRL(CADRL):
    def __init__(self):
        super().__init__()
        self._frags = None        # packed MFMA operand fragments (device) + the version they were packed at
        self._ws = None
        self._bufs = {}

    # ------------------------------------------------------------------ device plumbing
    def _gpu_device(self):
        d = self.device if isinstance(self.device, torch.device) else torch.device(self.device or "cuda")
        if d.type != "cuda":
            # the reference accepts --device cpu; this build's look-ahead only exists as HIP kernels
            d = torch.device("cuda", torch.cuda.current_device())
        return d

    def _packed(self, dev):
        raise NotImplementedError

    def _lookahead(self, st, E, N, dev, want_attention=False, env_next=None, epsilon=0.0):
        """Launch mcn_sarl_predict on an EnvState struct; returns (values[E,A], best[E], best_val[E], att); the chosen
        actions [E,2] (table row of `best`, zero where the robot stands on its goal) are left in self._bufs["action"].
        env_next = (next_hpos [E,N,2], next_hvel [E,N,2], rewards [E,A]): the `query_env` form -- the env's
        look-ahead states and rewards instead of propagate + compute_reward."""
        if self.action_space is None:
            raise RuntimeError("action space not built")
        A = len(self.action_space)
        key = (E, N, A, dev)
        if self._bufs.get("key") != key:
            nbytes = _hip.lib.mcn_sarl_workspace_bytes(E, N, A)
            self._bufs = {
                "key": key,
                "ws": torch.empty(nbytes // 4, dtype=torch.float32, device=dev),
                "values": torch.empty(E, A, dtype=torch.float64, device=dev),
                "best": torch.empty(E, dtype=torch.int32, device=dev),
                "best_val": torch.empty(E, dtype=torch.float64, device=dev),
                "action": torch.empty(E, 2, dtype=torch.float64, device=dev),
                "table": torch.from_numpy(np.ascontiguousarray(self._action_table)).to(dev),
                "att": None,
            }
        b = self._bufs
        if want_attention and b["att"] is None:
            b["att"] = torch.empty(E, A, N, dtype=torch.float32, device=dev)
        net = self._packed(dev)
        kin = _hip.KIN_UNICYCLE if self.kinematics == "unicycle" else _hip.KIN_HOLONOMIC
        gamma_pow = pow(self.gamma, self.time_step * self._v_pref)       # multi_human_rl.py:52
        npos, nvel, rew = env_next if env_next is not None else (None, None, None)
        rc = _hip.lib.mcn_sarl_predict(C.byref(net), st, _hip.ptr(b["table"]), A, float(self.time_step), gamma_pow, kin,
                                       _hip.ptr(b["ws"]), _hip.ptr(b["values"]), _hip.ptr(b["best"]),
                                       _hip.ptr(b["best_val"]), _hip.ptr(b["att"]) if want_attention else None,
                                       _hip.ptr(npos), _hip.ptr(nvel), _hip.ptr(rew), _hip.ptr(b["action"]),
                                       float(epsilon),
                                       # a fresh 63-bit seed per call from torch's host generator (no device launch):
                                       # torch.manual_seed() makes training rollouts reproducible
                                       int(torch.randint(0, 2 ** 62, (1,)).item()) if epsilon > 0 else 0,
                                       E, N, _hip.stream_ptr(dev))
        _hip.check(rc, "mcn_sarl_predict")
        return b["values"], b["best"], b["best_val"], b["att"]

    def _query_env(self, venv):
        """`query_env = true` (multi_human_rl.py:37-38): what `env.onestep_lookahead(action)` returns for every action
        of the table, for all E envs of the batched env `venv`.  The humans react to the robot's CURRENT state
        (crowd_sim.py:336-342), so their next states do not depend on the candidate action: ONE mcn_env_step(update = 0)
        gives them; the reward (swept-circle test against the candidate action, goal test, time limit) does, and comes
        from one given-velocity mcn_env_step over the E x A (env, action) pairs on a scratch copy of the state.
        Returns (next_hpos [E,N,2], next_hvel [E,N,2], rewards [E,A])."""
        E, N, dev = venv.num_envs, venv._alloc_N, venv.device
        A = len(self.action_space)
        table = self._bufs["table"] if self._bufs.get("table") is not None else \
            torch.from_numpy(np.ascontiguousarray(self._action_table)).to(dev)
        ob, _, _, _ = venv.onestep_lookahead(torch.zeros(E, 2, dtype=torch.float64, device=dev))
        npos, nvel = ob.pos.clone(), ob.vel.clone()
        rep = lambda t: t.repeat_interleave(A, 0).contiguous()
        x = dict(hpos=rep(venv.hpos), hvel=rep(venv.hvel), hrad=rep(venv.hrad), rpos=rep(venv.rpos), rvel=rep(venv.rvel),
                 rgoal=rep(venv.rgoal), rrad=rep(venv.rrad), rvpref=rep(venv.rvpref), rtheta=rep(venv.rtheta),
                 gtime=rep(venv.gtime))
        st = _hip.EnvState()
        for k, v in x.items():
            setattr(st, k, _hip.ptr(v))
        st.hgoal, st.hvpref = _hip.ptr(x["hpos"]), _hip.ptr(x["hrad"])      # not read with given velocities
        acts = table.repeat(E, 1).contiguous()
        given = rep(nvel)
        rec = torch.zeros(E * A, 3, dtype=torch.float64, device=dev)
        out = _hip.EnvOut(_hip.ptr(rec), None, None, None)
        cfg = venv._cfg_struct("given")
        cfg.count_hh = 0
        cfg.track_human_times = 0
        _hip.check(_hip.lib.mcn_env_step(cfg, st, _hip.ptr(acts), _hip.ptr(given), out, None, E * A, N, 1,
                                         _hip.stream_ptr(dev)), "mcn_env_step")
        return npos, nvel, rec[:, 0].reshape(E, A).contiguous()

    # ------------------------------------------------------------------ reference surface (E = 1)
    def predict(self, state):
        if self.phase is None or self.device is None:
            raise AttributeError("Phase, device attributes have to be set!")
        if self.phase == "train" and self.epsilon is None:
            raise AttributeError("Epsilon attribute has to be set in training phase")
        zero = ActionXY(0, 0) if self.kinematics == "holonomic" else ActionRot(0, 0)
        if self.reach_destination(state):
            return zero
        me, humans = state.self_state, state.human_states
        if self.action_space is None:
            self.build_action_space(me.v_pref)
        if self.with_om:
            raise NotImplementedError("occupancy maps (with_om) are outside this build's scope")
        probability = np.random.random()                       # drawn unconditionally, as the reference does
        if self.phase == "train" and probability < self.epsilon:
            max_action = self.action_space[np.random.choice(len(self.action_space))]
        else:
            dev = self._gpu_device()
            N = len(humans)
            # one host row, one host-to-device copy; the state arrays are views of it (16-byte aligned: pairs first)
            row = [c for h in humans for c in (h.px, h.py)] + [c for h in humans for c in (h.vx, h.vy)] + \
                  [me.px, me.py, me.vx, me.vy, me.gx, me.gy] + [h.radius for h in humans] + [me.radius, me.v_pref, me.theta]
            stage = torch.tensor(row, dtype=torch.float64).to(dev)
            names = ("hpos", "hvel", "rpos", "rvel", "rgoal", "hrad", "rrad", "rvpref", "rtheta")
            sizes = (2 * N, 2 * N, 2, 2, 2, N, 1, 1, 1)
            shapes = ((N, 2), (N, 2), (1, 2), (1, 2), (1, 2), (N,), (1,), (1,), (1,))
            bufs = {k: piece.view(shp) for k, piece, shp in zip(names, torch.split(stage, sizes), shapes)}
            st = _hip.EnvState()
            for k, v in bufs.items():
                setattr(st, k, _hip.ptr(v))
            self._v_pref = me.v_pref
            env_next = None
            if self.query_env:
                # the reference asks its env (whose internal state is the current one) 81 times; here the E = 1 view's
                # batched env answers for the whole table at once
                venv = self.env.__dict__.get("_vec") if hasattr(self.env, "__dict__") else None
                if venv is None:
                    raise AttributeError("query_env needs set_env(CrowdSim)")
                self.env._push_host_state()
                env_next = self._query_env(venv)
            values, best, _, att = self._lookahead(st, 1, N, dev, want_attention=True, env_next=env_next)
            vals = values[0].cpu().numpy()
            self.action_values = vals.tolist()
            idx = int(best.item())
            # the reference's model keeps the weights of its LAST forward, i.e. of the last candidate action
            # (sarl.py:56,88-89), and that is what env.step stores per step; the chosen action's are kept beside them
            a_host = att[0].cpu().numpy()
            self._last_attention = a_host[-1].copy()
            self.chosen_attention_weights = a_host[max(idx, 0)].copy()
            if idx < 0 or not np.isfinite(vals[idx]):
                # every value NaN <=> `value > max_value` never fired in the reference loop
                raise ValueError("Value network is not well trained. ")
            max_action = self.action_space[idx]
        if self.phase == "train":
            self.last_state = self.transform(state)
        return max_action

    def transform(self, state):
        """multi_human_rl.py:90-104: [N,13] rotated rows (float32) for the replay memory."""
        rows = torch.cat([torch.Tensor([state.self_state + h]).to(self.device) for h in state.human_states], dim=0)
        if self.with_om:
            raise NotImplementedError("occupancy maps (with_om) are outside this build's scope")
        return self.rotate(rows)

    def transform_batch(self, env):
        """`transform` for every env of a VecCrowdSim: [E,N,13] float32 rotated joint states (what the reference
        stores as `last_state` in train phase, multi_human_rl.py:60-61)."""
        E, N = env.num_envs, env._alloc_N
        f = torch.float32
        rob = torch.cat([env.rpos, env.rvel, env.rrad.unsqueeze(1), env.rgoal, env.rvpref.unsqueeze(1),
                         env.rtheta.unsqueeze(1)], 1).to(f)                                  # [E,9]
        hum = torch.cat([env.hpos, env.hvel, env.hrad.unsqueeze(2)], 2).to(f)                 # [E,N,5]
        rows = torch.cat([rob.unsqueeze(1).expand(E, N, 9), hum], 2).reshape(E * N, 14)
        return self.rotate(rows).view(E, N, 13)

    def input_dim(self):
        return self.joint_state_dim + (self.cell_num ** 2 * self.om_channel_size if self.with_om else 0)

    # ------------------------------------------------------------------ batched surface
    def predict_batch(self, env, want_values=False, hcount=None):
        """Look-ahead for all E environments of a VecCrowdSim: greedy in phase 'test' / 'val'; in phase 'train' each
        env independently takes a uniformly random table action with probability `epsilon` (multi_human_rl.py:27-29,
        one draw per env per step from torch's device generator) -- `best` is -2 for those envs.

        Returns (actions [E,2] float64 device tensor, best [E] int32; -1 where the robot already
        stands on its goal and the zero action is returned, multi_human_rl.py:22-23).  Both (and `values`) are the
        policy's own output buffers, written by the look-ahead launch: valid until the next predict_batch call.
        hcount ([E] int32 device tensor, optional): env e shows only its first hcount[e] pedestrians to the policy
        (the reference simply hands `predict` a shorter list, e.g. datagen.py:347-363)."""
        if self.action_space is None:
            self.build_action_space(float(env.robot.v_pref))
        dev = env.device
        self._v_pref = float(env.robot.v_pref)
        st = env._st
        if hcount is not None:
            if hcount.dtype != torch.int32 or not hcount.is_contiguous() or hcount.numel() != env.num_envs:
                raise ValueError("hcount must be a contiguous int32 tensor with one entry per env")
            st = _hip.EnvState.from_buffer_copy(env._st)
            st.hcount = _hip.ptr(hcount)
        env_next = None
        if self.query_env:
            if hcount is not None:
                raise NotImplementedError("query_env with per-env pedestrian counts")
            if self._bufs.get("table") is None:
                self._bufs["table"] = torch.from_numpy(np.ascontiguousarray(self._action_table)).to(dev)
            env_next = self._query_env(env)
        eps = float(getattr(self, "epsilon", 0) or 0) if self.phase == "train" else 0.0
        # epsilon-greedy happens inside the look-ahead's argmax kernel (one draw per env per step from a counter-based
        # stream seeded from torch's generator): best == -2 marks the envs that explored
        values, best, best_val, _ = self._lookahead(st, env.num_envs, env._alloc_N, dev, env_next=env_next, epsilon=eps)
        actions = self._bufs["action"]              # written by the argmax kernel: no torch launches
        if want_values:
            return actions, best, values
        return actions, best
