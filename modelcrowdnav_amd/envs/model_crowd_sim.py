"""ModelCrowdSim on MI355X: the env whose humans are advanced by a learned world model
(reference: crowd_sim/envs/model_crowd_sim.py).

Differences from CrowdSim that the reference has and this keeps:
  * humans move with velocities handed in (`step(new_v=...)`, :347,417) or produced by
    `sim_world` (:398-407); no ORCA, no human-human overlap check, no human_times (:347-441);
  * `set_current_state(obs, robot_info)` replays a recorded scene (:339-345): goals 0, theta 0;
  * `reset(no_random_gen=True)` creates blank humans (:288-290); the seeded branch does NOT reseed numpy
    (:296 is commented out) and the generators also draw an initial velocity (:183-192,225);
  * case_size['train'] is uint32max - 2000 (:68).

The step itself is the same fused kernel as CrowdSim (env_step.hip) in MCN_HUMANS_GIVEN mode.
"""
import numpy as np
import torch

from . import scenarios as S
from .crowd_sim import CrowdSim, VecCrowdSim, _UINT32_MAX
from .utils.human import Human
from .utils.state import ObservableState


class VecModelCrowdSim(VecCrowdSim):
    def __init__(self, num_envs, device=None):
        super().__init__(num_envs, device)
        self.count_hh = False
        self.track_human_times = False
        self.init_velocity = True
        self.human_policy_name = "given"
        self._side = None            # side stream + event of prefetch_world()
        self._prefetched = None

    def _case_sizes(self, config):
        return {"train": _UINT32_MAX - 2000, "val": config.getint("env", "val_size"),
                "test": config.getint("env", "test_size")}

    def step(self, actions, update=True, new_v=None, noise=None):
        """model_crowd_sim.py:347-441.  new_v: [E,N,2] velocities, or None to ask `sim_world`
        (a VecSGANWorld-style callable: positions [E,N,2] (+ noise) -> velocities [E,N,2])."""
        if new_v is None and self._prefetched is not None:
            new_v, ev = self._prefetched
            self._prefetched = None
            torch.cuda.current_stream(self.device).wait_event(ev)
        if new_v is None:
            if self.sim_world is None:
                raise AttributeError("sim_world has to be set when new_v is not given")
            new_v = self.sim_world(self.hpos, noise) if noise is not None else self.sim_world(self.hpos)
        return super().step(actions, update=update, given_v=new_v)

    def prefetch_world(self, noise=None):
        """Start the world model's prediction for the COMING step on a side stream, so that it overlaps the robot's
        policy (the value-network look-ahead reads the same state and does not need the prediction); the next
        `step()` without `new_v` picks the result up.  The humans' reaction does not depend on the robot's action
        (model_crowd_sim.py:398-407 feeds the world model the current human states only), so the order
        prefetch -> predict -> step gives exactly the values of predict -> step."""
        if self.sim_world is None:
            raise AttributeError("sim_world has to be set")
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
        cur = torch.cuda.current_stream(self.device)
        self._side.wait_stream(cur)          # after everything queued so far (the previous step, its readers)
        with torch.cuda.stream(self._side):
            v = self.sim_world(self.hpos, noise) if noise is not None else self.sim_world(self.hpos)
            ev = torch.cuda.Event()
            ev.record(self._side)
        self._prefetched = (v, ev)

    def onestep_lookahead(self, actions):
        return self.step(actions, update=False)

    def set_current_state(self, hpos, hvel, hradius, robot_pos=None, robot_goal=None):
        """Batched :339-345: positions/velocities/radii [E,N,(2)] tensors; goals become 0."""
        E, N = hpos.shape[0], hpos.shape[1]
        if self._alloc_N != N:
            self._allocate(N)
        self.human_num = N
        self.hpos.copy_(hpos); self.hvel.copy_(hvel)
        self.hgoal.zero_()
        self.hrad.copy_(hradius); self.hvpref.fill_(self._human_v_pref)
        rr = self.spec().robot_row()
        self.rpos.copy_(torch.tensor([rr[S.PX], rr[S.PY]], dtype=torch.float64, device=self.device).expand(E, 2)
                        if robot_pos is None else robot_pos)
        self.rgoal.copy_(torch.tensor([rr[S.GX], rr[S.GY]], dtype=torch.float64, device=self.device).expand(E, 2)
                         if robot_goal is None else robot_goal)
        self.rvel.zero_(); self.rtheta.fill_(np.pi / 2); self.gtime.zero_()
        self.rrad.fill_(float(self.robot.radius)); self.rvpref.fill_(float(self.robot.v_pref))


class ModelCrowdSim(CrowdSim):
    """E = 1 view with the reference's surface (see CrowdSim)."""
    _vec_cls = VecModelCrowdSim
    _tracks_human_times = False

    def reset(self, phase="test", test_case=None, no_random_gen=False):
        """model_crowd_sim.py:268-334."""
        v = self._vec
        if v is None or v.robot is None:
            raise AttributeError("robot has to be set!")
        assert phase in ["train", "val", "test"]
        if test_case is not None:
            v.case_counter[phase] = test_case
        self.global_time = 0
        multi = bool(v.robot.policy.multiagent_training)
        self.human_times = [0] * (v.human_num if (phase == "test" or multi) else 1)
        robot = v.robot
        robot.set(0, -v.circle_radius, 0, v.circle_radius, 0, 0, np.pi / 2)
        if no_random_gen:
            self.humans = [Human(v.config, "humans") for _ in range(v.human_num)]
            for h in self.humans:
                h.set(0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)
            scen = np.array([[h.px, h.py, h.gx, h.gy, h.vx, h.vy, h.theta, h.radius, h.v_pref] for h in self.humans])
        else:
            n, rule = v._phase_rule(phase)
            case = v.case_counter[phase]
            if case >= 0:
                scen = S.generate(v.spec(), np.random, n, rule)       # no reseed here, as in the reference (:296)
                v.case_counter[phase] = (case + 1) % v.case_size[phase]
            else:
                assert phase == "test"
                if case != -1:
                    raise NotImplementedError
                r, vp = v._human_radius, v._human_v_pref
                scen = np.array([[0, -6, 0, 5, 0, 0, np.pi / 2, r, vp], [-5, -5, -5, 5, 0, 0, np.pi / 2, r, vp],
                                 [5, -5, 5, 5, 0, 0, np.pi / 2, r, vp]], np.float64)
            self.humans = self._new_humans(scen)
        v.load_scenarios(np.asarray(scen, np.float64).reshape(1, -1, 9))
        for agent in [robot] + self.humans:
            agent.time_step = v.time_step
            if agent.policy is not None:
                agent.policy.time_step = v.time_step
        self.states = list()
        if hasattr(robot.policy, "action_values"):
            self.action_values = list()
        self.attention_weights = list() if hasattr(robot.policy, "get_attention_weights") else None
        return [h.get_observable_state() for h in self.humans]

    def set_current_state(self, obs, robot_info=None, phase="train"):
        """model_crowd_sim.py:339-345."""
        self._vec.human_num = len(obs)
        self.reset(phase, no_random_gen=True)
        if robot_info is not None:
            self._vec.robot.set(robot_info.px, robot_info.py, robot_info.gx, robot_info.gy, 0, 0, np.pi / 2)
        for h, ob in zip(self.humans, obs):
            h.set(ob.px, ob.py, 0, 0, ob.vx, ob.vy, 0)

    def step(self, action, update=True, new_v=None):
        """model_crowd_sim.py:347-441."""
        v = self._vec
        if new_v is None:
            current_s = [h.get_observable_state().getvalue() for h in self.humans]
            if v.sim_world is None:
                raise AttributeError("sim_world has to be set when new_v is not given")
            sw = v.sim_world
            from ..policy.world_model import SGANWorld
            if isinstance(sw, torch.nn.Module) and not isinstance(sw, SGANWorld):
                # MlpWorld / AttentionWorld style module (model_crowd_sim.py:404-407): one float32 row [1, 4N] in, one
                # row of 2N velocities out
                dev = getattr(self, "device", None)
                if dev is None:
                    prm = next(sw.parameters(), None)
                    dev = prm.device if prm is not None else "cpu"
                x = torch.Tensor([current_s]).to(dev)
                x = x.reshape(x.size(0), -1)
                with torch.no_grad():
                    out = sw(x)[0]
                new_v = torch.reshape(out, (len(self.humans), 2)).tolist()
            else:
                new_v = sw(current_s)                   # SGANWorld-style callable -> [N,2]
        gv = torch.tensor(np.asarray(new_v, np.float64).reshape(1, len(self.humans), 2), dtype=torch.float64,
                          device=v.device)
        self._pending_given = gv
        return super().step(action, update=update)

    # CrowdSim.step calls v.step(actions, update=...); route the velocities through
    def _vec_step(self, actions, update):
        return self._vec.step(actions, update=update, new_v=self._pending_given)
