"""ORCA policy object (reference: crowd_sim/envs/policy/orca.py:7-132).

For the env's own humans this class is only a parameter carrier: VecCrowdSim reads
neighbor_dist / max_neighbors / time_horizon / safety_space from it and the solve runs fused
inside env_step.hip.  `predict()` serves agents outside that kernel (the ORCA-driven robot of
`test.py --policy orca`): one mcn_orca_batch launch replaces the rvo2 simulator round trip.
"""
import numpy as np

from .policy import Policy
from ..utils.action import ActionXY


class ORCA(Policy):
    def __init__(self):
        super().__init__()
        self.name = "ORCA"
        self.trainable = False
        self.multiagent_training = None
        self.kinematics = "holonomic"
        self.safety_space = 0
        self.neighbor_dist = 10
        self.max_neighbors = 10
        self.time_horizon = 5
        self.time_horizon_obst = 5
        self.radius = 0.3
        self.max_speed = 1
        self.sim = None          # kept for attribute compatibility; no simulator object exists here
        self._bufs = None

    def configure(self, config):
        return

    def set_phase(self, phase):
        return

    def predict(self, state):
        import torch
        from ... import _hip
        me, others = state.self_state, state.human_states
        m = len(others)
        dev = torch.device("cuda", torch.cuda.current_device())
        # float32 conversion happens exactly where rvo2's Cython layer did it (orca.py:99-126)
        self_row = np.array([me.px, me.py, me.vx, me.vy, me.radius + 0.01 + self.safety_space, me.v_pref,
                             me.gx - me.px, me.gy - me.py], dtype=np.float64).astype(np.float32)
        oth = np.zeros((max(m, 1), 5), np.float32)
        for k, o in enumerate(others):
            oth[k] = np.array([o.px, o.py, o.vx, o.vy, o.radius + 0.01 + self.safety_space],
                              dtype=np.float64).astype(np.float32)
        # one host buffer, one copy: [self 8 | others 5 M | count (int32 bits) | 2 output slots]
        M = max(m, 1)
        host = np.zeros(8 + 5 * M + 3, np.float32)
        host[:8], host[8:8 + 5 * M] = self_row, oth.reshape(-1)
        host[8 + 5 * M:9 + 5 * M].view(np.int32)[0] = m
        stage = torch.from_numpy(host).to(dev)
        d_self, d_oth = stage[:8], stage[8:8 + 5 * M]
        d_n, d_out = stage[8 + 5 * M:9 + 5 * M].view(torch.int32), stage[9 + 5 * M:]
        _hip.check(_hip.lib.mcn_orca_batch(_hip.ptr(d_self), _hip.ptr(d_oth), _hip.ptr(d_n), _hip.ptr(d_out),
                                           1, max(m, 1), float(self.neighbor_dist), int(self.max_neighbors),
                                           float(self.time_horizon), float(self.time_step), _hip.stream_ptr(dev)),
                   "mcn_orca_batch")
        v = d_out.cpu().numpy()
        self.last_state = state
        return ActionXY(float(v[0]), float(v[1]))

    def predict_batch(self, env):
        """ORCA-driven robot for all E envs of a VecCrowdSim (the imitation-learning demonstrator, train.py:150-160):
        one mcn_orca_batch launch with the robot as agent 0 and every human as a candidate neighbour."""
        import torch
        from ... import _hip
        E, N, dev = env.num_envs, env._alloc_N, env.device
        f = torch.float32
        extra = 0.01 + float(self.safety_space)
        me = torch.cat([env.rpos.to(f), env.rvel.to(f), (env.rrad + extra).to(f).unsqueeze(1),
                        env.rvpref.to(f).unsqueeze(1), (env.rgoal - env.rpos).to(f)], 1).contiguous()      # [E,8]
        oth = torch.cat([env.hpos.to(f), env.hvel.to(f), (env.hrad + extra).to(f).unsqueeze(2)], 2).contiguous()
        n = torch.full((E,), N, dtype=torch.int32, device=dev)
        out = torch.empty(E, 2, dtype=f, device=dev)
        _hip.check(_hip.lib.mcn_orca_batch(_hip.ptr(me), _hip.ptr(oth), _hip.ptr(n), _hip.ptr(out), E, N,
                                           float(self.neighbor_dist), int(self.max_neighbors),
                                           float(self.time_horizon), float(env.time_step), _hip.stream_ptr(dev)),
                   "mcn_orca_batch")
        return out.double(), None
