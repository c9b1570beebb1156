"""CrowdSim on MI355X: E environments resident in HBM, stepped by one fused HIP kernel.

Reference surface kept (crowd_sim/envs/crowd_sim.py): configure :58, set_robot :91,
reset :261, step :331, onestep_lookahead :325, plus the attributes its callers read
(case_size, case_counter, human_num, time_step, time_limit, global_time, humans, robot, states,
sim_world, device, test_sim), get_human_times :219 (an all-ORCA simulation to the end: one mcn_orca_batch
launch per simulated step).  render is host-only visual tooling and out of scope.

  VecCrowdSim  tensors in / tensors out, E envs, SoA float64 state (layout: include/mcn.h)
  CrowdSim     the E = 1 gym-style view returning the reference's value types
"""
import logging

import numpy as np
import torch

from .. import _hip
from . import scenarios as S
from .policy.policy_factory import policy_factory
from .utils import info as I
from .utils.human import Human
from .utils.state import ObservableState

_UINT32_MAX = int(np.iinfo(np.uint32).max)


class ObsBatch(object):
    """Observation of a batch: views of the state arrays (never copied by step()).

    pos, vel: [E,N,2] float64; radius: [E,N] float64.
    `tensor()` materialises the reference's per-human 5-tuple layout [E,N,5]
    (px,py,vx,vy,radius: crowd_sim/envs/utils/state.py:27-33) when a caller wants it.
    """

    def __init__(self, pos, vel, radius):
        self.pos, self.vel, self.radius = pos, vel, radius

    def tensor(self):
        return torch.cat([self.pos, self.vel, self.radius.unsqueeze(-1)], dim=-1)

    def to_states(self, e=0):
        t = self.tensor()[e].cpu().tolist()
        return [ObservableState(*row) for row in t]


class VecCrowdSim(object):
    def __init__(self, num_envs, device=None):
        self.num_envs = int(num_envs)
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.time_limit = self.time_step = None
        self.robot = None
        self.success_reward = self.collision_penalty = None
        self.discomfort_dist = self.discomfort_penalty_factor = None
        self.config = None
        self.case_capacity = self.case_size = self.case_counter = None
        self.randomize_attributes = None
        self.train_val_sim = self.test_sim = None
        self.square_width = self.circle_radius = None
        self.human_num = None
        self.look_ahead_in_sim = False
        self.sim_world = None
        self._sim_adapter = None
        self.human_policy_name = "orca"
        self._orca = policy_factory["orca"]()      # parameter carrier (orca.py:59-66)
        self._human_radius = self._human_v_pref = None
        self._alloc_N = None
        self._roll = None
        self._keep = []       # tensors referenced by the rollout struct
        self.count_hh = True
        # first-arrival bookkeeping (crowd_sim.py:418-421) and the humans' chosen velocities are extra HBM
        # traffic nobody on the rollout path reads: opt-in (the E = 1 CrowdSim view turns both on)
        self.track_human_times = False
        self.export_human_actions = False
        self.init_velocity = False     # ModelCrowdSim flavour sets True

    # ---------------------------------------------------------------- configuration
    def configure(self, config):
        """crowd_sim.py:58-89.  `look_ahead_in_sim` is optional here: the reference's shipped
        env.config lacks it although configure() requires it."""
        self.config = config
        self.time_limit = config.getint("env", "time_limit")
        self.time_step = config.getfloat("env", "time_step")
        self.randomize_attributes = config.getboolean("env", "randomize_attributes")
        self.success_reward = config.getfloat("reward", "success_reward")
        self.collision_penalty = config.getfloat("reward", "collision_penalty")
        self.discomfort_dist = config.getfloat("reward", "discomfort_dist")
        self.discomfort_penalty_factor = config.getfloat("reward", "discomfort_penalty_factor")
        if config.get("humans", "policy") != "orca":
            raise NotImplementedError
        self.case_capacity = {"train": _UINT32_MAX - 2000, "val": 1000, "test": 1000}
        self.case_size = self._case_sizes(config)
        self.train_val_sim = config.get("sim", "train_val_sim")
        self.test_sim = config.get("sim", "test_sim")
        self.square_width = config.getfloat("sim", "square_width")
        self.circle_radius = config.getfloat("sim", "circle_radius")
        self.human_num = config.getint("sim", "human_num")
        self.case_counter = {"train": 0, "test": 0, "val": 0}
        self.look_ahead_in_sim = config.getboolean("env", "look_ahead_in_sim", fallback=False)
        self._human_radius = config.getfloat("humans", "radius")
        self._human_v_pref = config.getfloat("humans", "v_pref")
        logging.info("human number: %d", self.human_num)
        logging.info("%s human's radius and preferred speed",
                     "Randomize" if self.randomize_attributes else "Not randomize")
        logging.info("Training simulation: %s, test simulation: %s", self.train_val_sim, self.test_sim)
        logging.info("Square width: %s, circle width: %s", self.square_width, self.circle_radius)

    def _case_sizes(self, config):
        # crowd_sim.py:71 hard-codes 100 training seeds
        return {"train": 100, "val": config.getint("env", "val_size"), "test": config.getint("env", "test_size")}

    def set_robot(self, robot):
        self.robot = robot

    # ---------------------------------------------------------------- device state
    def _allocate(self, N):
        E, dev, f64 = self.num_envs, self.device, torch.float64
        z = lambda *shape, dtype=f64: torch.zeros(*shape, dtype=dtype, device=dev)
        self.hpos, self.hvel, self.hgoal = z(E, N, 2), z(E, N, 2), z(E, N, 2)
        self.hrad, self.hvpref = z(E, N), z(E, N)
        self.rpos, self.rvel, self.rgoal = z(E, 2), z(E, 2), z(E, 2)
        self.rrad, self.rvpref = z(E), z(E)
        self.rtheta, self.gtime = z(E), z(E)
        self.human_times = z(E, N)
        # per-step outputs are one 24-byte record per env (mcn_step_rec); the named fields are strided views of it
        self.step_rec = z(E, 3)
        v = _hip.step_rec_views(self.step_rec)
        self.reward, self.dmin, self.done, self.info, self.hh_count = (v[k] for k in
                                                                       ("reward", "dmin", "done", "info", "hh_count"))
        self.human_act = z(E, N, 2)
        self.nobs_pos, self.nobs_vel = z(E, N, 2), z(E, N, 2)
        self._alloc_N = N
        self._st = _hip.EnvState(*[_hip.ptr(t) for t in (self.hpos, self.hvel, self.hgoal, self.hrad, self.hvpref,
                                                         self.rpos, self.rvel, self.rgoal, self.rrad, self.rvpref,
                                                         self.rtheta, self.gtime, self.human_times)])
        # work queue of the deferred 3-D LP (mcn.h: mcn_env_out.lp3_queue): zero-filled once, then the kernels' own
        # (the library parks 3-D LPs only for crowds of >= 8 ORCA neighbours in batches of >= 16 384 wavefronts, or when
        #  mcn_tuning.lp3_defer forces it: small queues are always provided, large ones only where they will be used)
        nq = int(_hip.lib.mcn_env_lp3_queue_bytes(E, N))
        wanted = nq > 0 and (nq <= (256 << 20) or _hip.get_tuning().lp3_defer == 1 or
                             (N >= 8 and -(-E // (64 // N)) >= 16384))
        self.lp3_queue = torch.zeros(nq, dtype=torch.uint8, device=dev) if wanted else None
        self._out = _hip.EnvOut(*[_hip.ptr(t) for t in (self.step_rec, self.human_act, self.nobs_pos, self.nobs_vel,
                                                        self.lp3_queue)])
        self._out_lean = _hip.EnvOut(*[_hip.ptr(t) for t in (self.step_rec, None, self.nobs_pos, self.nobs_vel,
                                                             self.lp3_queue)])

    def _cfg_struct(self, human_policy=None):
        hp = {"orca": _hip.HUMANS_ORCA, "linear": _hip.HUMANS_LINEAR, "given": _hip.HUMANS_GIVEN}[
            human_policy or self.human_policy_name]
        kin = _hip.KIN_UNICYCLE if getattr(self.robot, "kinematics", "holonomic") == "unicycle" else _hip.KIN_HOLONOMIC
        o = self._orca
        return _hip.EnvCfg(self.time_step, float(self.time_limit), self.success_reward, self.collision_penalty,
                           self.discomfort_dist, self.discomfort_penalty_factor, float(o.safety_space),
                           float(o.neighbor_dist), float(o.time_horizon), int(o.max_neighbors),
                           1 if self.robot.visible else 0, hp, kin,
                           1 if self.count_hh else 0, 1 if self.track_human_times else 0)

    def spec(self):
        return S.ScenarioSpec(self.circle_radius, self.square_width, self.discomfort_dist, self._human_radius,
                              self._human_v_pref, self.robot.radius, self.randomize_attributes, self.init_velocity)

    def load_scenarios(self, scen, robot_rows=None):
        """Upload host scenarios [E,N,9] (scenarios.py column order) and reset clocks."""
        scen = np.asarray(scen, np.float64)
        E, N = scen.shape[0], scen.shape[1]
        if E != self.num_envs:
            raise ValueError("expected %d scenarios, got %d" % (self.num_envs, E))
        if self._alloc_N != N:
            self._allocate(N)
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        self.hpos.copy_(up(scen[:, :, [S.PX, S.PY]]))
        self.hgoal.copy_(up(scen[:, :, [S.GX, S.GY]]))
        self.hvel.copy_(up(scen[:, :, [S.VX, S.VY]]))
        self.hrad.copy_(up(scen[:, :, S.RAD]))
        self.hvpref.copy_(up(scen[:, :, S.VPREF]))
        if robot_rows is None:
            robot_rows = np.tile(self.spec().robot_row(), (E, 1))      # crowd_sim.py:284
        robot_rows = np.asarray(robot_rows, np.float64)
        self.rpos.copy_(up(robot_rows[:, [S.PX, S.PY]]))
        self.rgoal.copy_(up(robot_rows[:, [S.GX, S.GY]]))
        self.rvel.copy_(up(robot_rows[:, [S.VX, S.VY]]))
        self.rrad.copy_(up(robot_rows[:, S.RAD]))
        self.rvpref.fill_(float(self.robot.v_pref))
        self.rtheta.copy_(up(robot_rows[:, S.TH]))
        self.gtime.zero_()
        self.human_times.zero_()
        self.human_num = N

    def device_pool(self, seed, first_case, count, human_num=None, rule="circle_crossing"):
        """`count` scenarios generated ON the device (mcn_scenario_pool): the reference's placement rules with a
        counter-based random stream keyed by (seed, case id) -- statistically equivalent to CrowdSim.reset, not
        bit-identical to it (scenarios.py keeps the bit-exact host generator).  Returns the pool tensors that
        load_device_scenarios / attach_rollout(pool=...) accept."""
        N = int(human_num or self.human_num)
        sp, dev = self.spec(), self.device
        rr = sp.robot_row()
        cfg = _hip.ScenarioCfg(sp.circle_radius, sp.square_width, sp.discomfort_dist, sp.human_radius, sp.human_v_pref,
                               sp.robot_radius, (rr[S.PX], rr[S.PY]), (rr[S.GX], rr[S.GY]),
                               {"circle_crossing": _hip.RULE_CIRCLE, "square_crossing": _hip.RULE_SQUARE}[rule],
                               1 if sp.randomize_attributes else 0)
        z = lambda *shape: torch.zeros(*shape, dtype=torch.float64, device=dev)
        pool = dict(hpos=z(count, N, 2), hgoal=z(count, N, 2), hrad=z(count, N), hvpref=z(count, N))
        rc = _hip.lib.mcn_scenario_pool(cfg, int(seed) & 0xFFFFFFFFFFFFFFFF, int(first_case), int(count), N,
                                        _hip.ptr(pool["hpos"]), _hip.ptr(pool["hgoal"]), _hip.ptr(pool["hrad"]),
                                        _hip.ptr(pool["hvpref"]), _hip.stream_ptr(dev))
        _hip.check(rc, "mcn_scenario_pool")
        return pool

    def load_device_scenarios(self, pool, rows):
        """Start every env from pool row rows[e] (device pool of device_pool()); robot at its start pose."""
        rows = torch.as_tensor(rows, device=self.device, dtype=torch.long)
        E, N = self.num_envs, pool["hpos"].shape[1]
        if rows.numel() != E:
            raise ValueError("expected %d rows, got %d" % (E, rows.numel()))
        if self._alloc_N != N:
            self._allocate(N)
        self.hpos.copy_(pool["hpos"][rows]); self.hgoal.copy_(pool["hgoal"][rows]); self.hvel.zero_()
        self.hrad.copy_(pool["hrad"][rows]); self.hvpref.copy_(pool["hvpref"][rows])
        rr = torch.tensor(self.spec().robot_row(), dtype=torch.float64, device=self.device)
        self.rpos.copy_(rr[[S.PX, S.PY]].expand(E, 2)); self.rgoal.copy_(rr[[S.GX, S.GY]].expand(E, 2))
        self.rvel.zero_(); self.rrad.fill_(float(rr[S.RAD])); self.rvpref.fill_(float(self.robot.v_pref))
        self.rtheta.fill_(float(rr[S.TH])); self.gtime.zero_(); self.human_times.zero_()
        self.human_num = N

    # ---------------------------------------------------------------- gym surface (batched)
    def _phase_rule(self, phase):
        multi = bool(self.robot.policy.multiagent_training)
        if not multi:
            self.train_val_sim = "circle_crossing"      # crowd_sim.py:276-277
        if phase in ("train", "val"):
            return (self.human_num if multi else 1), self.train_val_sim
        return self.human_num, self.test_sim

    def reset(self, phase="test", test_cases=None):
        """crowd_sim.py:261-323 for every env.  Env e takes case counter+e unless test_cases is given."""
        if self.robot is None:
            raise AttributeError("robot has to be set!")
        assert phase in ["train", "val", "test"]
        E = self.num_envs
        if test_cases is None:
            base = self.case_counter[phase]
            cases = [(base + e) % self.case_size[phase] for e in range(E)]
        else:
            cases = [int(c) for c in (test_cases if hasattr(test_cases, "__len__") else [test_cases] * E)]
        n, rule = self._phase_rule(phase)
        scen = S.scenario_pool(self.spec(), phase, cases, n, rule)
        self.load_scenarios(scen)
        self.case_counter[phase] = (cases[-1] + 1) % self.case_size[phase]
        return self.observation()

    def observation(self):
        return ObsBatch(self.hpos, self.hvel, self.hrad)

    def step(self, actions, update=True, given_v=None):
        """crowd_sim.py:331-434 for every env: one mcn_env_step launch, no host sync.

        actions: [E,2] float64 device tensor ((vx,vy) holonomic, (v,r) unicycle).
        Returns (ObsBatch, reward[E] f64, done[E] u8, info[E] u8 codes) -- all device views that
        the next step overwrites.  With update=False the ObsBatch holds the look-ahead states.
        """
        E, N = self.num_envs, self._alloc_N
        if actions.dtype != torch.float64 or not actions.is_contiguous() or tuple(actions.shape) != (E, 2):
            actions = actions.to(self.device, torch.float64).reshape(E, 2).contiguous()
        policy = None
        if given_v is not None:
            policy = "given"
            if given_v.dtype != torch.float64 or not given_v.is_contiguous():
                given_v = given_v.to(self.device, torch.float64).contiguous()
            if tuple(given_v.shape) != (E, N, 2):
                raise ValueError("given_v must be [E,N,2]")
        cfg = self._cfg_struct(policy)
        rc = _hip.lib.mcn_env_step(cfg, self._st, _hip.ptr(actions), _hip.ptr(given_v),
                                   self._out if self.export_human_actions else self._out_lean,
                                   self._roll if (self._roll is not None and update) else None,
                                   E, N, 1 if update else 0, _hip.stream_ptr(self.device))
        _hip.check(rc, "mcn_env_step")
        if update:
            ob = self.observation()
        else:
            ob = ObsBatch(self.nobs_pos, self.nobs_vel, self.hrad)
        return ob, self.reward, self.done, self.info

    def rollout(self, actions):
        """T consecutive step(update=True) calls for an action sequence known up front (actions: [T,E,2] float64
        device tensor) -- the step loop of Explorer.run_k_episodes (explorer.py:69-99) for a robot that does not look
        at the observation.  One mcn_env_rollout call: for small crowds the T steps are a single launch with the
        state held in registers.  Results equal T step() calls bit for bit; the returned views are those of step T.
        """
        E, N = self.num_envs, self._alloc_N
        if actions.dim() != 3 or tuple(actions.shape[1:]) != (E, 2):
            raise ValueError("actions must be [T,E,2]")
        if actions.dtype != torch.float64 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(self.device, torch.float64).contiguous()
        rc = _hip.lib.mcn_env_rollout(self._cfg_struct(None), self._st, _hip.ptr(actions), int(actions.shape[0]),
                                      self._out if self.export_human_actions else self._out_lean,
                                      self._roll, E, N, _hip.stream_ptr(self.device))
        _hip.check(rc, "mcn_env_rollout")
        return self.observation(), self.reward, self.done, self.info

    def onestep_lookahead(self, actions):
        """crowd_sim.py:325-329: a non-mutating step; with look_ahead_in_sim the humans' next states come from the
        learned world model instead of ORCA (step_in_sim, crowd_sim.py:633-696)."""
        if self.look_ahead_in_sim:
            return self.step_in_sim(actions)
        return self.step(actions, update=False)

    def step_in_sim(self, actions):
        """crowd_sim.py:633-696: same collision / reward ladder (no human-human check), next human states =
        position + sim_world velocity * dt.  `sim_world` is a VecSGANWorld / VecTorchWorld style callable."""
        if self.sim_world is None:
            raise AttributeError("sim_world has to be set for look_ahead_in_sim")
        sw = self.sim_world
        if isinstance(sw, torch.nn.Module):
            # the reference hands a bare MlpWorld / AttentionWorld module over (crowd_sim.py:684-688): evaluate it with
            # its tensor convention for all E scenes; through torch, so that a module under training is always current
            if self._sim_adapter is None or self._sim_adapter[0] is not sw:
                from ..policy.world_model import VecTorchWorld
                self._sim_adapter = (sw, VecTorchWorld(sw, self))
            sw = self._sim_adapter[1]
        new_v = sw(self.hpos)
        keep = self.count_hh
        self.count_hh = False
        try:
            return self.step(actions, update=False, given_v=new_v)
        finally:
            self.count_hh = keep

    # ---------------------------------------------------------------- fused rollout bookkeeping
    def attach_rollout(self, gamma, pool=None, case_stride=1, first_cases=None, fin_slots=1, danger_episodes=0,
                       danger_short_from=0):
        """Enable Explorer-style return accounting (explorer.py:124) and, when `pool` ([P,N,9]
        host scenarios) is given, in-kernel auto-reset from that HBM-resident pool.  `danger_episodes` > 0: the
        "too close" counters (explorer.py:88-90) only cover each env's first that many episodes (one fewer for envs
        from index `danger_short_from - 1` on, when that is > 0: the partial last round of k episodes over E envs)."""
        E, dev = self.num_envs, self.device
        horizon = int(round(self.time_limit / self.time_step)) + 2
        v_pref = float(self.robot.v_pref)
        disc = np.array([pow(gamma, t * self.time_step * v_pref) for t in range(horizon)], np.float64)
        t = {}
        t["disc"] = torch.from_numpy(disc).to(dev)
        # per-env rollout state is one 32-byte record (mcn_roll_rec); the named entries are strided views of it
        t["state"] = torch.zeros(E, 4, dtype=torch.float64, device=dev)
        t.update(_hip.roll_rec_views(t["state"]))
        t["fin_return"] = torch.zeros(fin_slots, E, dtype=torch.float64, device=dev)
        t["fin_time"] = torch.zeros(fin_slots, E, dtype=torch.float64, device=dev)
        t["fin_info"] = torch.zeros(fin_slots, E, dtype=torch.uint8, device=dev)
        r = _hip.Rollout()
        r.disc_table, r.disc_len, r.fin_slots = _hip.ptr(t["disc"]), horizon, int(fin_slots)
        r.danger_episodes, r.danger_short_from = int(danger_episodes), int(danger_short_from)
        for k in ("state", "fin_return", "fin_time", "fin_info"):
            setattr(r, k, _hip.ptr(t[k]))
        if pool is not None:
            if isinstance(pool, dict):                   # device pool (device_pool()): used in place, zero velocities
                P, N = pool["hpos"].shape[0], pool["hpos"].shape[1]
                if N != self._alloc_N:
                    raise ValueError("pool N %d != env N %d" % (N, self._alloc_N))
                t["pool_hpos"], t["pool_hgoal"] = pool["hpos"], pool["hgoal"]
                t["pool_hrad"], t["pool_hvpref"] = pool["hrad"], pool["hvpref"]
                t["pool_hvel"] = None
            else:
                pool = np.asarray(pool, np.float64)
                P, N = pool.shape[0], pool.shape[1]
                if N != self._alloc_N:
                    raise ValueError("pool N %d != env N %d" % (N, self._alloc_N))
                up = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
                t["pool_hpos"] = up(pool[:, :, [S.PX, S.PY]]); t["pool_hgoal"] = up(pool[:, :, [S.GX, S.GY]])
                t["pool_hrad"] = up(pool[:, :, S.RAD]); t["pool_hvpref"] = up(pool[:, :, S.VPREF])
                t["pool_hvel"] = up(pool[:, :, [S.VX, S.VY]])
            nc = np.arange(E) % P if first_cases is None else np.asarray(first_cases) % P
            t["next_case"].copy_(torch.from_numpy(nc.astype(np.int32)).to(dev))
            r.pool_hpos, r.pool_hgoal = _hip.ptr(t["pool_hpos"]), _hip.ptr(t["pool_hgoal"])
            r.pool_hrad, r.pool_hvpref = _hip.ptr(t["pool_hrad"]), _hip.ptr(t["pool_hvpref"])
            r.pool_hvel = _hip.ptr(t["pool_hvel"])
            r.pool_size, r.case_stride = P, int(case_stride) % P
            rr = self.spec().robot_row()
            r.robot_start[0], r.robot_start[1] = rr[S.PX], rr[S.PY]
            r.robot_goal[0], r.robot_goal[1] = rr[S.GX], rr[S.GY]
            r.robot_theta0 = rr[S.TH]
        self._roll, self.rollout_buffers = r, t
        return t

    def detach_rollout(self):
        self._roll, self.rollout_buffers = None, None


class CrowdSim(object):
    """E = 1 view with the reference's gym surface and value types.

    Works for `gym.make('CrowdSim-v0')`-style callers: Explorer.run_k_episodes
    (crowd_nav/utils/explorer.py:54,69), the policies' onestep_lookahead (cadrl.py:159,
    multi_human_rl.py:38) and the drivers (test.py:64-72,90-95).  Every step is one
    mcn_env_step launch plus a small device->host read-back that refreshes `humans` / `robot`.
    """
    metadata = {"render.modes": ["human"]}
    _vec_cls = VecCrowdSim
    _tracks_human_times = True

    def __init__(self, device=None):
        self._vec = None
        self._device = device
        self.humans = None
        self.global_time = None
        self.human_times = None
        self.states = None
        self.action_values = None
        self.attention_weights = None
        self.device = None

    # attributes the reference keeps on the env object live on the vector env
    _FORWARD = ("time_limit", "time_step", "robot", "success_reward", "collision_penalty", "discomfort_dist",
                "discomfort_penalty_factor", "config", "case_capacity", "case_size", "case_counter",
                "randomize_attributes", "train_val_sim", "test_sim", "square_width", "circle_radius",
                "human_num", "look_ahead_in_sim", "sim_world")

    def __getattr__(self, name):
        if name in CrowdSim._FORWARD:
            v = self.__dict__.get("_vec")
            return None if v is None else getattr(v, name)
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in CrowdSim._FORWARD and self.__dict__.get("_vec") is not None:
            setattr(self._vec, name, value)
        else:
            object.__setattr__(self, name, value)

    def _ensure_vec(self):
        if self._vec is None:
            object.__setattr__(self, "_vec", self._vec_cls(1, self._device))
        return self._vec

    def configure(self, config):
        self._ensure_vec().configure(config)

    def set_robot(self, robot):
        self._ensure_vec().set_robot(robot)

    # ------------------------------------------------------------------------------
    def _new_humans(self, scen):
        v = self._vec
        humans = []
        for row in scen:
            h = Human(v.config, "humans")
            h.set(row[S.PX], row[S.PY], row[S.GX], row[S.GY], row[S.VX], row[S.VY], row[S.TH],
                  radius=row[S.RAD], v_pref=row[S.VPREF])
            humans.append(h)
        return humans

    def reset(self, phase="test", test_case=None):
        """crowd_sim.py:261-323: seeds numpy's GLOBAL stream like the reference, so policies that
        draw from it afterwards (epsilon-greedy, multi_human_rl.py:28-30) see the same numbers."""
        v = self._vec
        if v is None or v.robot is None:
            raise AttributeError("robot has to be set!")
        assert phase in ["train", "val", "test"]
        if test_case is not None:
            v.case_counter[phase] = test_case
        self.global_time = 0
        multi = bool(v.robot.policy.multiagent_training)
        self.human_times = [0] * (v.human_num if (phase == "test" or multi) else 1)
        n, rule = v._phase_rule(phase)
        robot = v.robot
        robot.set(0, -v.circle_radius, 0, v.circle_radius, 0, 0, np.pi / 2)
        case = v.case_counter[phase]
        if case >= 0:
            np.random.seed(S.SEED_OFFSET[phase] + case)
            scen = S.generate(v.spec(), np.random, n, rule)
            v.case_counter[phase] = (case + 1) % v.case_size[phase]
        else:
            assert phase == "test"
            if case != -1:
                raise NotImplementedError
            r, vp = v._human_radius, v._human_v_pref      # crowd_sim.py:297-303 debug layout
            scen = np.array([[0, -6, 0, 5, 0, 0, np.pi / 2, r, vp], [-5, -5, -5, 5, 0, 0, np.pi / 2, r, vp],
                             [5, -5, 5, 5, 0, 0, np.pi / 2, r, vp]], np.float64)
        self.humans = self._new_humans(scen)
        v.load_scenarios(scen[None])
        for agent in [robot] + self.humans:
            agent.time_step = v.time_step
            if agent.policy is not None:
                agent.policy.time_step = v.time_step
        self.states = list()
        if hasattr(robot.policy, "action_values"):
            self.action_values = list()
        if hasattr(robot.policy, "get_attention_weights"):
            self.attention_weights = list()
        if robot.sensor != "coordinates":
            raise NotImplementedError
        return [h.get_observable_state() for h in self.humans]

    def _push_host_state(self):
        """Host mirrors are authoritative between steps in the E = 1 view (callers may call
        robot.set()/human.set()): re-upload the few scalars before launching."""
        v, r = self._vec, self._vec.robot
        hs = self.humans
        n = len(hs)
        # one host row -> one host-to-device copy -> one multi-tensor device copy into the state arrays
        row = [c for h in hs for c in (h.px, h.py)] + [c for h in hs for c in (h.vx, h.vy)] + \
              [c for h in hs for c in (h.gx, h.gy)] + [h.radius for h in hs] + [h.v_pref for h in hs] + \
              [r.px, r.py, r.vx, r.vy, r.gx, r.gy, r.radius, r.v_pref, r.theta, self.global_time]
        dsts = [v.hpos, v.hvel, v.hgoal, v.hrad, v.hvpref, v.rpos, v.rvel, v.rgoal, v.rrad, v.rvpref, v.rtheta, v.gtime]
        sizes = [2 * n, 2 * n, 2 * n, n, n, 2, 2, 2, 1, 1, 1, 1]
        if len(self.human_times) == n:
            row += list(self.human_times)
            dsts.append(v.human_times)
            sizes.append(n)
        stage = torch.tensor(row, dtype=torch.float64).to(v.device)
        srcs = [piece.view(d.shape) for piece, d in zip(torch.split(stage, sizes), dsts)]
        if hasattr(torch, "_foreach_copy_"):
            torch._foreach_copy_(dsts, srcs)
        else:
            for d, piece in zip(dsts, srcs):
                d.copy_(piece)

    def _action_tensor(self, action):
        return torch.tensor([[action[0], action[1]]], dtype=torch.float64, device=self._vec.device)

    def onestep_lookahead(self, action):
        """crowd_sim.py:325-329."""
        if not self.look_ahead_in_sim:
            return self.step(action, update=False)
        return self.step_in_sim(action)

    def step_in_sim(self, action):
        """crowd_sim.py:633-696: the look-ahead whose humans are moved by the learned world model instead of ORCA --
        same swept test / reward ladder, next human states = position + sim_world velocity * dt, nothing mutated."""
        if self._vec.sim_world is None:
            raise AttributeError("sim_world has to be set for look_ahead_in_sim")
        self._in_sim = True
        try:
            return self.step(action, update=False)
        finally:
            self._in_sim = False

    def step(self, action, update=True):
        """crowd_sim.py:331-434."""
        v = self._vec
        robot = v.robot
        self._push_host_state()
        track = self._tracks_human_times and len(self.human_times) == len(self.humans)
        v.track_human_times = track
        v.export_human_actions = True
        ob, reward, done, info = self._vec_step(self._action_tensor(action), update)
        # everything the host mirrors need, gathered on the device and fetched with ONE synchronising copy
        n = len(self.humans)
        host = torch.cat([ob.pos[0].reshape(-1), ob.vel[0].reshape(-1), v.step_rec[0], v.rpos[0], v.rvel[0],
                          v.rtheta, v.gtime, v.human_times[0]]).cpu().numpy()
        pos, vel = host[:2 * n].reshape(n, 2).tolist(), host[2 * n:4 * n].reshape(n, 2).tolist()
        rec = host[4 * n:4 * n + 3]
        flags = rec[2:3].view(np.uint8)                              # mcn_step_rec: done, info at bytes 16, 17
        reward, done, code = float(rec[0]), bool(flags[0]), int(flags[1])
        info_obj = I.from_code(code, float(rec[1]))
        tail = host[4 * n + 3:]                                      # rpos 2, rvel 2, rtheta, gtime, human_times n
        if update:
            self.states.append([robot.get_full_state(), [h.get_full_state() for h in self.humans]])
            if hasattr(robot.policy, "action_values"):
                self.action_values.append(robot.policy.action_values)
            if hasattr(robot.policy, "get_attention_weights"):
                self.attention_weights.append(robot.policy.get_attention_weights())
            robot.px, robot.py, robot.vx, robot.vy = float(tail[0]), float(tail[1]), float(tail[2]), float(tail[3])
            if robot.kinematics == "unicycle":
                robot.theta = float(tail[4])
            for h, p, w in zip(self.humans, pos, vel):
                h.px, h.py, h.vx, h.vy = p[0], p[1], w[0], w[1]
            self.global_time = float(tail[5])
            if track:
                self.human_times = tail[6:6 + n].tolist()
            out = [h.get_observable_state() for h in self.humans]
        else:
            out = [ObservableState(p[0], p[1], w[0], w[1], h.radius) for h, p, w in zip(self.humans, pos, vel)]
        return out, reward, done, info_obj

    def _vec_step(self, actions, update):
        if getattr(self, "_in_sim", False):
            return self._vec.step_in_sim(actions)
        return self._vec.step(actions, update=update)

    def get_human_times(self, max_steps=8000):
        """crowd_sim.py:219-258: once the robot has arrived, run everybody (robot = agent 0, then the humans) to the end
        in ONE centralised ORCA simulation and return each human's first-arrival time.  The reference drives an
        rvo2.PyRVOSimulator(time_step, 10, 10, 5, 5, 0.3, 1) with agents at their own radius / v_pref; here every
        simulated step is one mcn_orca_batch launch (B = N + 1 agents, each with the other N as candidates in index
        order) and the float32 position update rvo2 does inside doStep (velocity = newVelocity; position += velocity *
        timeStep) is done on the host copies in float32.  ORCA parity vs rvo2 is unpinned (oracle/mcn_oracle.c); rvo2
        enumerates neighbours in kd-tree order, which only matters for exactly equal distances.
        The reference loops for ever if somebody never arrives (it only logs past t = 1000); this stops after
        `max_steps` simulated steps with the same warning."""
        v, robot = self._vec, self._vec.robot
        if not robot.reached_destination():
            raise ValueError("Episode is not done yet")
        agents = [robot] + self.humans
        B, f32, dev = len(agents), np.float32, v.device
        M = B - 1
        pos = np.array([a.get_position() for a in agents], np.float64).astype(f32)
        vel = np.array([a.get_velocity() for a in agents], np.float64).astype(f32)
        rad = np.array([a.radius for a in agents], np.float64).astype(f32)
        vmax = np.array([a.v_pref for a in agents], np.float64).astype(f32)
        dt32 = f32(v.time_step)
        others_idx = np.array([[j for j in range(B) if j != i] for i in range(B)], np.int64).reshape(B, M)
        d_n = torch.full((B,), M, dtype=torch.int32, device=dev)
        d_out = torch.empty(B, 2, dtype=torch.float32, device=dev)
        max_time, steps = 1000, 0
        while not all(self.human_times):
            pref = np.zeros((B, 2), np.float64)
            for i, agent in enumerate(agents):
                vel_pref = np.array(agent.get_goal_position()) - np.array(agent.get_position())
                if np.linalg.norm(vel_pref) > 1:
                    vel_pref /= np.linalg.norm(vel_pref)
                pref[i] = vel_pref
            me = np.concatenate([pos, vel, rad[:, None], vmax[:, None], pref.astype(f32)], 1).astype(f32)       # [B,8]
            oth = np.concatenate([pos[others_idx], vel[others_idx], rad[others_idx][..., None]], 2).astype(f32)  # [B,M,5]
            d_me, d_oth = torch.from_numpy(me).to(dev), torch.from_numpy(np.ascontiguousarray(oth)).to(dev)
            _hip.check(_hip.lib.mcn_orca_batch(_hip.ptr(d_me), _hip.ptr(d_oth), _hip.ptr(d_n), _hip.ptr(d_out), B, max(M, 1),
                                               10.0, 10, 5.0, float(v.time_step), _hip.stream_ptr(dev)), "mcn_orca_batch")
            vel = d_out.cpu().numpy().astype(f32)
            pos = (pos + vel * dt32).astype(f32)
            self.global_time += v.time_step
            steps += 1
            if self.global_time > max_time:
                logging.warning("Simulation cannot terminate!")
            for i, human in enumerate(self.humans):
                if self.human_times[i] == 0 and human.reached_destination():
                    self.human_times[i] = self.global_time
            robot.set_position((float(pos[0, 0]), float(pos[0, 1])))
            for i, human in enumerate(self.humans):
                human.set_position((float(pos[i + 1, 0]), float(pos[i + 1, 1])))
            self.states.append([robot.get_full_state(), [h.get_full_state() for h in self.humans]])
            if steps >= max_steps:
                break
        return self.human_times

    def render(self, mode="human", output_file=None, render_weight=False):
        raise NotImplementedError("render is matplotlib host tooling, out of scope for this build (SURVEY.md 2)")
