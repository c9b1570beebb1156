"""ORACLE -- test infrastructure only.  ctypes binding of oracle/libmcn_oracle.so.

Build with `make -C oracle` (also done by __graft_entry__.build()).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmcn_oracle.so")

INFO_NOTHING, INFO_DANGER, INFO_REACHGOAL, INFO_COLLISION, INFO_TIMEOUT = range(5)
HUMANS_ORCA, HUMANS_LINEAR, HUMANS_GIVEN = range(3)


class Cfg(C.Structure):
    _fields_ = [
        ("time_step", C.c_double), ("time_limit", C.c_double),
        ("success_reward", C.c_double), ("collision_penalty", C.c_double),
        ("discomfort_dist", C.c_double), ("discomfort_penalty_factor", C.c_double),
        ("robot_visible", C.c_int), ("human_policy", C.c_int),
        ("count_hh", C.c_int), ("track_human_times", C.c_int),
        ("orca_safety_space", C.c_double),
        ("orca_neighbor_dist", C.c_float), ("orca_max_neighbors", C.c_int),
        ("orca_time_horizon", C.c_float), ("orca_max_speed", C.c_float),
        ("robot_unicycle", C.c_int),
    ]


def default_cfg(**kw):
    """env.config:1-13,33 + orca.py:60-66 defaults."""
    c = Cfg(0.25, 25.0, 1.0, -0.25, 0.2, 0.5, 0, HUMANS_ORCA, 1, 1, 0.0, 10.0, 10, 5.0, 1.0, 0)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


_lib = None


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "mcn_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-B"], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.mcn_oracle_point_to_segment_dist.restype = C.c_double
        _lib.mcn_oracle_point_to_segment_dist.argtypes = [C.c_double] * 6
        _lib.mcn_oracle_orca_lines.restype = C.c_int
    return _lib


def _p(a, t):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(t))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def lp3_entries(reset=True):
    """Agents that fell through to linearProgram3 since the last reset (coverage assertions)."""
    f = lib().mcn_oracle_lp3_entries
    f.restype = C.c_long
    return int(f(C.c_int(1 if reset else 0)))


def point_to_segment_dist(x1, y1, x2, y2, x3, y3):
    return lib().mcn_oracle_point_to_segment_dist(x1, y1, x2, y2, x3, y3)


def orca_agent(pos, vel, radius, max_speed, pref, opos, ovel, orad,
               neighbor_dist=10.0, max_neighbors=10, time_horizon=5.0, time_step=0.25):
    """One agent's ORCA velocity (float32).  opos/ovel: [M,2], orad: [M]."""
    opos = _f32(opos).reshape(-1, 2); ovel = _f32(ovel).reshape(-1, 2); orad = _f32(orad).reshape(-1)
    m = opos.shape[0]
    opx, opy = _f32(opos[:, 0]), _f32(opos[:, 1])
    ovx, ovy = _f32(ovel[:, 0]), _f32(ovel[:, 1])
    ox, oy = C.c_float(), C.c_float()
    f = C.c_float
    lib().mcn_oracle_orca_agent(f(pos[0]), f(pos[1]), f(vel[0]), f(vel[1]), f(radius), f(max_speed),
                                f(pref[0]), f(pref[1]), C.c_int(m),
                                _p(opx, f), _p(opy, f), _p(ovx, f), _p(ovy, f), _p(orad, f),
                                f(neighbor_dist), C.c_int(max_neighbors), f(time_horizon), f(time_step),
                                C.byref(ox), C.byref(oy))
    return np.float32(ox.value), np.float32(oy.value)


def orca_lines(pos, vel, radius, opos, ovel, orad,
               neighbor_dist=10.0, max_neighbors=10, time_horizon=5.0, time_step=0.25):
    opos = _f32(opos).reshape(-1, 2); ovel = _f32(ovel).reshape(-1, 2); orad = _f32(orad).reshape(-1)
    m = opos.shape[0]
    opx, opy = _f32(opos[:, 0]), _f32(opos[:, 1])
    ovx, ovy = _f32(ovel[:, 0]), _f32(ovel[:, 1])
    out = np.zeros((max(m, 1), 4), np.float32)
    f = C.c_float
    n = lib().mcn_oracle_orca_lines(f(pos[0]), f(pos[1]), f(vel[0]), f(vel[1]), f(radius), C.c_int(m),
                                    _p(opx, f), _p(opy, f), _p(ovx, f), _p(ovy, f), _p(orad, f),
                                    f(neighbor_dist), C.c_int(max_neighbors), f(time_horizon), f(time_step),
                                    _p(out, f))
    return out[:n]


class EnvState:
    """Flat float64 batch state, [E] / [E,N] arrays (row-major)."""
    FIELDS_H = ("hpx", "hpy", "hvx", "hvy", "hgx", "hgy", "hr", "hvpref")
    FIELDS_R = ("rpx", "rpy", "rvx", "rvy", "rgx", "rgy", "rr")

    def __init__(self, E, N):
        self.E, self.N = E, N
        for k in self.FIELDS_H:
            setattr(self, k, np.zeros((E, N)))
        for k in self.FIELDS_R:
            setattr(self, k, np.zeros(E))
        self.gtime = np.zeros(E)
        self.rtheta = np.zeros(E)
        self.human_times = np.zeros((E, N))

    def copy(self):
        o = EnvState(self.E, self.N)
        for k in self.FIELDS_H + self.FIELDS_R + ("gtime", "rtheta", "human_times"):
            setattr(o, k, getattr(self, k).copy())
        return o


def env_step(cfg, st, ax, ay, update=True, given_v=None):
    """Batched env step.  Returns dict(reward, done, info, dmin, hh_count, human_act[, nobs_*])."""
    E, N = st.E, st.N
    d = C.c_double
    ax = np.ascontiguousarray(ax, np.float64); ay = np.ascontiguousarray(ay, np.float64)
    reward = np.zeros(E); done = np.zeros(E, np.uint8); info = np.zeros(E, np.uint8)
    dmin = np.zeros(E); hh = np.zeros(E, np.int32)
    nobs = [np.zeros((E, N)) for _ in range(4)]
    hact = np.zeros((E, N, 2))
    gv = None if given_v is None else np.ascontiguousarray(given_v, np.float64)
    lib().mcn_oracle_env_step(C.byref(cfg), C.c_int(E), C.c_int(N), C.c_int(1 if update else 0),
                              _p(st.hpx, d), _p(st.hpy, d), _p(st.hvx, d), _p(st.hvy, d),
                              _p(st.hgx, d), _p(st.hgy, d), _p(st.hr, d), _p(st.hvpref, d),
                              _p(st.rpx, d), _p(st.rpy, d), _p(st.rvx, d), _p(st.rvy, d),
                              _p(st.rgx, d), _p(st.rgy, d), _p(st.rr, d), _p(st.rtheta, d),
                              _p(st.gtime, d), _p(st.human_times, d),
                              _p(ax, d), _p(ay, d), _p(gv, d),
                              _p(reward, d), _p(done, C.c_uint8), _p(info, C.c_uint8), _p(dmin, d),
                              _p(hh, C.c_int32),
                              _p(nobs[0], d), _p(nobs[1], d), _p(nobs[2], d), _p(nobs[3], d),
                              _p(hact, d))
    out = dict(reward=reward, done=done, info=info, dmin=dmin, hh_count=hh, human_act=hact)
    if not update:
        out.update(nobs_px=nobs[0], nobs_py=nobs[1], nobs_vx=nobs[2], nobs_vy=nobs[3])
    return out


def lookahead_reward(st, actions, dt):
    """MultiHumanRL.compute_reward over [E] x [A] candidate actions -> [E,A]."""
    E, N = st.E, st.N
    act = np.ascontiguousarray(actions, np.float64).reshape(-1, 2)
    A = act.shape[0]
    out = np.zeros((E, A))
    d = C.c_double
    lib().mcn_oracle_lookahead_reward(C.c_int(E), C.c_int(N), C.c_int(A), d(dt),
                                      _p(st.rpx, d), _p(st.rpy, d), _p(st.rgx, d), _p(st.rgy, d), _p(st.rr, d),
                                      _p(st.hpx, d), _p(st.hpy, d), _p(st.hvx, d), _p(st.hvy, d), _p(st.hr, d),
                                      _p(act, d), _p(out, d))
    return out
