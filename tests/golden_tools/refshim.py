"""Golden-vector tooling (this container only; never shipped to the GPU box as a dependency).

Makes `/root/reference` importable by placing throw-away modules in
sys.modules for the packages the image lacks:

  gym       -- `Env` base class + `envs.registration.register` no-op
               (crowd_sim/__init__.py:1, crowd_sim/envs/crowd_sim.py:2,15)
  attrdict  -- dict with attribute access (crowd_nav/policy/world_model.py:4,109)
  pykalman  -- empty module: imported at the top of the vendored trajnetplusplustools/kalman.py, never used by the
               ndjson reader that crowd_nav/utils/misc.py:GetRealData drives
  rvo2      -- `PyRVOSimulator` with exactly the methods the reference calls
               (crowd_sim/envs/policy/orca.py:95-129, crowd_sim/envs/crowd_sim.py:231-255),
               backed by THIS REPO's C ORCA restatement (oracle/mcn_oracle.c).

The rvo2 shim does not make ORCA parity pinned: fixtures produced through it pin the
reference's *plumbing around* the solver (which neighbours, which radii, which
preferred velocity, float32 round trips, integration), not the solver itself.
"""
import sys
import types

import numpy as np

REFERENCE_ROOT = "/root/reference"


class _PyRVOSimulator:
    def __init__(self, timeStep, neighborDist, maxNeighbors, timeHorizon, timeHorizonObst, radius=0.0, maxSpeed=0.0,
                 velocity=(0.0, 0.0)):
        from oracle import cport
        self._c = cport
        self.time_step = np.float32(timeStep)
        self.pos, self.vel, self.pref, self.newvel = [], [], [], []
        self.params = []   # (neighborDist, maxNeighbors, timeHorizon, radius, maxSpeed)

    def addAgent(self, pos, neighborDist, maxNeighbors, timeHorizon, timeHorizonObst, radius, maxSpeed,
                 velocity=(0.0, 0.0)):
        self.pos.append(np.asarray(pos, dtype=np.float32).copy())
        self.vel.append(np.asarray(velocity, dtype=np.float32).copy())
        self.pref.append(np.zeros(2, np.float32))
        self.newvel.append(np.zeros(2, np.float32))
        self.params.append((np.float32(neighborDist), int(maxNeighbors), np.float32(timeHorizon),
                            np.float32(radius), np.float32(maxSpeed)))
        return len(self.pos) - 1

    def getNumAgents(self):
        return len(self.pos)

    def setAgentPosition(self, i, p):
        self.pos[i] = np.asarray(p, dtype=np.float32).copy()

    def setAgentVelocity(self, i, v):
        self.vel[i] = np.asarray(v, dtype=np.float32).copy()

    def setAgentPrefVelocity(self, i, v):
        self.pref[i] = np.asarray(v, dtype=np.float32).copy()

    def getAgentVelocity(self, i):
        return (float(self.vel[i][0]), float(self.vel[i][1]))

    def getAgentPosition(self, i):
        return (float(self.pos[i][0]), float(self.pos[i][1]))

    def doStep(self):
        n = len(self.pos)
        for i in range(n):
            nd, mn, th, r, ms = self.params[i]
            others = [j for j in range(n) if j != i]
            opos = np.array([self.pos[j] for j in others], np.float32).reshape(-1, 2)
            ovel = np.array([self.vel[j] for j in others], np.float32).reshape(-1, 2)
            orad = np.array([self.params[j][3] for j in others], np.float32)
            vx, vy = self._c.orca_agent(self.pos[i], self.vel[i], r, ms, self.pref[i], opos, ovel, orad,
                                        neighbor_dist=nd, max_neighbors=mn, time_horizon=th,
                                        time_step=self.time_step)
            self.newvel[i] = np.array([vx, vy], np.float32)
        for i in range(n):
            self.vel[i] = self.newvel[i].copy()
            self.pos[i] = (self.pos[i] + self.vel[i] * self.time_step).astype(np.float32)


def install():
    """Insert the shim modules and put the reference on sys.path.  Idempotent."""
    sys.dont_write_bytecode = True
    if "gym" not in sys.modules:
        gym = types.ModuleType("gym")

        class Env(object):
            pass
        gym.Env = Env
        envs = types.ModuleType("gym.envs")
        reg = types.ModuleType("gym.envs.registration")
        reg.register = lambda **kw: None
        gym.envs = envs
        envs.registration = reg
        sys.modules.update({"gym": gym, "gym.envs": envs, "gym.envs.registration": reg})
    if "attrdict" not in sys.modules:
        ad = types.ModuleType("attrdict")

        class AttrDict(dict):
            __getattr__ = dict.__getitem__
        ad.AttrDict = AttrDict
        sys.modules["attrdict"] = ad
    if "rvo2" not in sys.modules:
        rvo2 = types.ModuleType("rvo2")
        rvo2.PyRVOSimulator = _PyRVOSimulator
        sys.modules["rvo2"] = rvo2
    if "pykalman" not in sys.modules:
        # trajnetplusplustools/__init__.py:10 pulls in kalman.py, whose only top-level use of pykalman is the import;
        # nothing on the ingest path (reader.py, data.py) touches it
        try:
            import pykalman  # noqa: F401
        except ImportError:
            sys.modules["pykalman"] = types.ModuleType("pykalman")
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
