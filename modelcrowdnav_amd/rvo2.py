"""`rvo2`-shaped module: the reference's one native boundary, served by the HIP ORCA kernel.

The reference imports Python-RVO2 (`import rvo2`, crowd_sim/envs/policy/orca.py:2, crowd_sim/envs/crowd_sim.py:5) and
drives `rvo2.PyRVOSimulator` through exactly these calls:

    PyRVOSimulator(timeStep, neighborDist, maxNeighbors, timeHorizon, timeHorizonObst, radius, maxSpeed)
    addAgent(pos, neighborDist, maxNeighbors, timeHorizon, timeHorizonObst, radius, maxSpeed, velocity) -> int
    setAgentPosition / setAgentVelocity / setAgentPrefVelocity(i, (x, y));  doStep()
    getAgentVelocity(i) / getAgentPosition(i) -> (x, y);  getNumAgents()
                                    (orca.py:95-129, crowd_sim.py:231-255, model_crowd_sim.py:238-265)

This class keeps that surface -- same names, argument order and meaning, Python floats in and out, float32 inside as
in the Cython layer -- so that `sys.modules["rvo2"] = modelcrowdnav_amd.rvo2` lets the reference's own CrowdSim / ORCA
classes run unchanged on an MI355X.  `doStep()` is one `mcn_orca_batch` launch for all agents (one per
distinct (neighborDist, maxNeighbors, timeHorizon) group if agents were added with different ones), followed by RVO2's
update `velocity = newVelocity; position += velocity * timeStep` in float32 on the host copies.

It is the compatibility path (one launch + one device round trip per doStep: tens of microseconds for a handful of
agents); the fast path for many environments is VecCrowdSim / mcn_env_step, which fuses this solve into the env step.
There is no CPU fallback: without the HIP library or a GPU, doStep() raises.

ORCA parity vs the real rvo2 is UNPINNED (rvo2 is absent from the reference tree and from this image):
the solver follows the published algorithm with RVO2 v2.0.x conventions; obstacles are not implemented (the reference
never adds any).  rvo2 enumerates neighbours in kd-tree order, this kernel in agent-index order: that can only matter
for exactly equal distances.
"""
import numpy as np

_MAX_CANDIDATES = 32          # MCN_MAX_HUMANS: candidate neighbours per agent and launch (include/mcn.h)
_MAX_NEIGHBORS = 10           # MCN_MAX_LINES


class PyRVOSimulator(object):
    def __init__(self, timeStep, neighborDist, maxNeighbors, timeHorizon, timeHorizonObst, radius=0.0, maxSpeed=0.0,
                 velocity=(0.0, 0.0)):
        self._dt = float(timeStep)
        self._defaults = (float(neighborDist), int(maxNeighbors), float(timeHorizon), float(timeHorizonObst),
                          float(radius), float(maxSpeed), (float(velocity[0]), float(velocity[1])))
        self._pos, self._vel, self._pref = [], [], []
        self._param = []                # (neighborDist, maxNeighbors, timeHorizon) per agent
        self._rad, self._vmax = [], []
        self._time = 0.0

    # ------------------------------------------------------------------ agents
    def addAgent(self, pos, neighborDist=None, maxNeighbors=None, timeHorizon=None, timeHorizonObst=None, radius=None,
                 maxSpeed=None, velocity=None):
        d = self._defaults
        nd = d[0] if neighborDist is None else float(neighborDist)
        mn = d[1] if maxNeighbors is None else int(maxNeighbors)
        th = d[2] if timeHorizon is None else float(timeHorizon)
        if mn < 0 or mn > _MAX_NEIGHBORS:
            raise ValueError("maxNeighbors must be in 0..%d (MCN_MAX_LINES), got %d" % (_MAX_NEIGHBORS, mn))
        if not th > 0:
            raise ValueError("timeHorizon must be positive")
        vel = d[6] if velocity is None else velocity
        self._pos.append(np.array([pos[0], pos[1]], np.float64).astype(np.float32))
        self._vel.append(np.array([vel[0], vel[1]], np.float64).astype(np.float32))
        self._pref.append(np.zeros(2, np.float32))
        self._param.append((np.float32(nd), mn, np.float32(th)))
        self._rad.append(np.float32(d[4] if radius is None else radius))
        self._vmax.append(np.float32(d[5] if maxSpeed is None else maxSpeed))
        return len(self._pos) - 1

    def getNumAgents(self):
        return len(self._pos)

    def setAgentPosition(self, i, p):
        self._pos[i] = np.array([p[0], p[1]], np.float64).astype(np.float32)

    def setAgentVelocity(self, i, v):
        self._vel[i] = np.array([v[0], v[1]], np.float64).astype(np.float32)

    def setAgentPrefVelocity(self, i, v):
        self._pref[i] = np.array([v[0], v[1]], np.float64).astype(np.float32)

    def setAgentRadius(self, i, r):
        self._rad[i] = np.float32(r)

    def setAgentMaxSpeed(self, i, s):
        self._vmax[i] = np.float32(s)

    def getAgentPosition(self, i):
        return (float(self._pos[i][0]), float(self._pos[i][1]))

    def getAgentVelocity(self, i):
        return (float(self._vel[i][0]), float(self._vel[i][1]))

    def getAgentPrefVelocity(self, i):
        return (float(self._pref[i][0]), float(self._pref[i][1]))

    def getAgentRadius(self, i):
        return float(self._rad[i])

    def getAgentMaxSpeed(self, i):
        return float(self._vmax[i])

    def getTimeStep(self):
        return self._dt

    def setTimeStep(self, dt):
        self._dt = float(dt)

    def getGlobalTime(self):
        return self._time

    # ------------------------------------------------------------------ obstacles: none in this project
    def addObstacle(self, vertices):
        raise NotImplementedError("obstacles are not implemented: the reference never adds any (orca.py:95-129)")

    def processObstacles(self):
        raise NotImplementedError("obstacles are not implemented: the reference never adds any (orca.py:95-129)")

    # ------------------------------------------------------------------ the step
    def _candidates(self, pos):
        """[n][M] candidate indices per agent in agent-index order.  With more than 32 other agents only the 32
        nearest are handed to the kernel, which then keeps at most maxNeighbors <= 10 of them: the same set."""
        n = len(pos)
        m = min(n - 1, _MAX_CANDIDATES)
        idx = np.zeros((n, max(m, 1)), np.int64)
        if n - 1 <= _MAX_CANDIDATES:
            for i in range(n):
                idx[i, :m] = [j for j in range(n) if j != i]
            return idx, m
        d2 = ((pos[:, None, :].astype(np.float64) - pos[None, :, :].astype(np.float64)) ** 2).sum(2)
        np.fill_diagonal(d2, np.inf)
        near = np.argsort(d2, axis=1, kind="stable")[:, :m]
        idx[:, :m] = np.sort(near, axis=1)
        return idx, m

    def doStep(self):
        import torch
        from . import _hip
        n = len(self._pos)
        if n == 0:
            self._time += self._dt
            return
        if not torch.cuda.is_available():
            raise RuntimeError("modelcrowdnav_amd.rvo2 needs a GPU: doStep() runs mcn_orca_batch (no CPU fallback)")
        dev = torch.device("cuda", torch.cuda.current_device())
        pos, vel, pref = np.stack(self._pos), np.stack(self._vel), np.stack(self._pref)
        rad, vmax = np.array(self._rad, np.float32), np.array(self._vmax, np.float32)
        idx, m = self._candidates(pos)
        me = np.concatenate([pos, vel, rad[:, None], vmax[:, None], pref], 1).astype(np.float32)              # [n,8]
        oth = np.concatenate([pos[idx], vel[idx], rad[idx][..., None]], 2).astype(np.float32)                # [n,M,5]
        new_vel = np.zeros((n, 2), np.float32)
        groups = {}
        for i, prm in enumerate(self._param):
            groups.setdefault(prm, []).append(i)
        for (nd, mn, th), rows in groups.items():
            r = np.array(rows, np.int64)
            d_me = torch.from_numpy(np.ascontiguousarray(me[r])).to(dev)
            d_oth = torch.from_numpy(np.ascontiguousarray(oth[r])).to(dev)
            d_n = torch.full((len(rows),), m, dtype=torch.int32, device=dev)
            d_out = torch.empty(len(rows), 2, dtype=torch.float32, device=dev)
            _hip.check(_hip.lib.mcn_orca_batch(_hip.ptr(d_me), _hip.ptr(d_oth), _hip.ptr(d_n), _hip.ptr(d_out),
                                               len(rows), max(m, 1), float(nd), int(mn), float(th), float(self._dt),
                                               _hip.stream_ptr(dev)), "mcn_orca_batch")
            new_vel[r] = d_out.cpu().numpy()
        dt32 = np.float32(self._dt)
        for i in range(n):
            self._vel[i] = new_vel[i].copy()
            self._pos[i] = (self._pos[i] + new_vel[i] * dt32).astype(np.float32)
        self._time += self._dt
