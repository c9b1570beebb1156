"""ctypes binding of libmcn_hip.so (C ABI: include/mcn.h).

The library is the product: there is no CPU fallback.  If it is missing, importing this
module raises, and every env / policy entry point above it fails with it.
"""
import ctypes as C
import os

import torch  # noqa: F401  -- FIRST: libmcn_hip.so must bind to the HIP runtime torch already loaded,
#                     otherwise two runtimes coexist and torch streams are meaningless to our launches

_HERE = os.path.dirname(os.path.abspath(__file__))
# MCN_HIP_LIB: an explicit path to another build of the same library (the diagnostic build of
# `make -C modelcrowdnav_amd/csrc diag`, used by tools/fixed_cost.py); version() says which build is loaded
LIB_PATH = os.environ.get("MCN_HIP_LIB") or os.path.join(_HERE, "csrc", "libmcn_hip.so")

MCN_OK, MCN_EINVAL, MCN_ELAUNCH = 0, -1, -2
MAX_HUMANS, MAX_LINES = 32, 10
INFO_NOTHING, INFO_DANGER, INFO_REACHGOAL, INFO_COLLISION, INFO_TIMEOUT = range(5)
HUMANS_ORCA, HUMANS_LINEAR, HUMANS_GIVEN = range(3)
KIN_HOLONOMIC, KIN_UNICYCLE = 0, 1

_vp, _d, _f, _i = C.c_void_p, C.c_double, C.c_float, C.c_int32


class EnvCfg(C.Structure):
    _fields_ = [
        ("time_step", _d), ("time_limit", _d), ("success_reward", _d), ("collision_penalty", _d),
        ("discomfort_dist", _d), ("discomfort_penalty_factor", _d), ("orca_safety_space", _d),
        ("orca_neighbor_dist", _f), ("orca_time_horizon", _f),
        ("orca_max_neighbors", _i), ("robot_visible", _i), ("human_policy", _i),
        ("robot_kinematics", _i), ("count_hh", _i), ("track_human_times", _i),
    ]


class EnvState(C.Structure):
    _fields_ = [(k, _vp) for k in ("hpos", "hvel", "hgoal", "hrad", "hvpref", "rpos", "rvel", "rgoal", "rrad", "rvpref",
                                   "rtheta", "gtime", "human_times", "hcount")]


class EnvOut(C.Structure):
    _fields_ = [(k, _vp) for k in ("rec", "human_act", "nobs_pos", "nobs_vel", "lp3_queue")]


class StepRec(C.Structure):
    """mcn_step_rec: what one env reports per step (24 bytes)."""
    _fields_ = [("reward", _d), ("dmin", _d), ("done", C.c_uint8), ("info", C.c_uint8), ("reserved", C.c_uint16),
                ("hh_count", _i)]


class RollRec(C.Structure):
    """mcn_roll_rec: rollout state of one env (32 bytes)."""
    _fields_ = [("ep_return", _d), ("ep_steps", _i), ("fin_count", _i), ("next_case", _i), ("danger_count", _i),
                ("danger_dist_sum", _d)]


def step_rec_views(rec):
    """Typed strided views of a [E,3] float64 tensor laid out as mcn_step_rec[E]."""
    import torch
    b, w = rec.view(torch.uint8), rec.view(torch.int32)
    return dict(reward=rec[:, 0], dmin=rec[:, 1], done=b[:, 16], info=b[:, 17], hh_count=w[:, 5])


def roll_rec_views(state):
    """Typed strided views of a [E,4] float64 tensor laid out as mcn_roll_rec[E]."""
    import torch
    w = state.view(torch.int32)
    return dict(ep_return=state[:, 0], ep_steps=w[:, 2], fin_count=w[:, 3], next_case=w[:, 4],
                danger_count=w[:, 5], danger_dist_sum=state[:, 3])


class Rollout(C.Structure):
    _fields_ = [
        ("disc_table", _vp), ("disc_len", _i), ("danger_episodes", _i),
        ("state", _vp),
        ("fin_return", _vp), ("fin_time", _vp), ("fin_info", _vp), ("fin_slots", _i), ("danger_short_from", _i),
        ("pool_hpos", _vp), ("pool_hgoal", _vp), ("pool_hrad", _vp), ("pool_hvpref", _vp), ("pool_hvel", _vp),
        ("pool_size", _i), ("case_stride", _i),
        ("robot_start", _d * 2), ("robot_goal", _d * 2), ("robot_theta0", _d),
    ]


class ScenarioCfg(C.Structure):
    _fields_ = [("circle_radius", _d), ("square_width", _d), ("discomfort_dist", _d), ("human_radius", _d),
                ("human_v_pref", _d), ("robot_radius", _d), ("robot_start", _d * 2), ("robot_goal", _d * 2),
                ("rule", _i), ("randomize_attributes", _i)]


RULE_CIRCLE, RULE_SQUARE = 0, 1


class Tuning(C.Structure):
    """mcn_tuning: dispatch overrides, -1 = automatic."""
    _fields_ = [(k, _i) for k in ("force_generic", "quad_max_envs", "quad_split", "rollout_fused", "rollout_split",
                                  "step_block", "diag_noop", "pair_stream", "lp3_defer", "sarl_x3")]


class McnError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "modelcrowdnav_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C modelcrowdnav_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.mcn_version.restype = C.c_char_p
    lib.mcn_env_step.argtypes = [C.POINTER(EnvCfg), C.POINTER(EnvState), _vp, _vp, C.POINTER(EnvOut),
                                 C.POINTER(Rollout), _i, _i, _i, _vp]
    lib.mcn_env_step.restype = C.c_int
    lib.mcn_env_lp3_queue_bytes.argtypes = [_i, _i]
    lib.mcn_env_lp3_queue_bytes.restype = C.c_int64
    lib.mcn_env_rollout.argtypes = [C.POINTER(EnvCfg), C.POINTER(EnvState), _vp, _i, C.POINTER(EnvOut),
                                    C.POINTER(Rollout), _i, _i, _vp]
    lib.mcn_env_rollout.restype = C.c_int
    lib.mcn_set_tuning.argtypes = [C.POINTER(Tuning)]
    lib.mcn_set_tuning.restype = C.c_int
    lib.mcn_get_tuning.argtypes = [C.POINTER(Tuning)]
    lib.mcn_get_tuning.restype = C.c_int
    lib.mcn_scenario_pool.argtypes = [C.POINTER(ScenarioCfg), C.c_uint64, C.c_int64, _i, _i, _vp, _vp, _vp, _vp, _vp]
    lib.mcn_scenario_pool.restype = C.c_int
    lib.mcn_orca_batch.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _f, _i, _f, _f, _vp]
    lib.mcn_orca_batch.restype = C.c_int
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int32)
    lib.mcn_pack_linear.argtypes = [fp, fp, _i, _i, ip, _i, ip, _i, fp, fp]
    lib.mcn_pack_linear.restype = C.c_int
    lib.mcn_sarl_workspace_bytes.argtypes = [_i, _i, _i]
    lib.mcn_sarl_workspace_bytes.restype = C.c_int64
    lib.mcn_sarl_lookahead.argtypes = [_vp, C.POINTER(EnvState), _vp, _i, _d, _d, _i, _vp, _vp, _vp, _vp, _vp,
                                       _i, _i, _vp]
    lib.mcn_sarl_lookahead.restype = C.c_int
    lib.mcn_sarl_lookahead_env.argtypes = [_vp, C.POINTER(EnvState), _vp, _i, _d, _d, _i, _vp, _vp, _vp, _vp, _vp,
                                           _vp, _vp, _vp, _i, _i, _vp]
    lib.mcn_sarl_lookahead_env.restype = C.c_int
    lib.mcn_sarl_predict.argtypes = [_vp, C.POINTER(EnvState), _vp, _i, _d, _d, _i, _vp, _vp, _vp, _vp, _vp,
                                     _vp, _vp, _vp, _vp, _d, C.c_uint64, _i, _i, _vp]
    lib.mcn_sarl_predict.restype = C.c_int
    lib.mcn_mlp_world_step.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _vp]
    lib.mcn_mlp_world_step.restype = C.c_int
    lib.mcn_attn_world_workspace_bytes.argtypes = [_i, _i]
    lib.mcn_attn_world_workspace_bytes.restype = C.c_int64
    lib.mcn_attn_world_step.argtypes = [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp]
    lib.mcn_attn_world_step.restype = C.c_int
    lib.mcn_sgan_workspace_bytes.argtypes = [_i, _i]
    lib.mcn_sgan_workspace_bytes.restype = C.c_int64
    lib.mcn_sgan_step.argtypes = [_vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _d, _i, _i, _vp]
    lib.mcn_sgan_step.restype = C.c_int
    lib.mcn_pack_x3_bytes.argtypes, lib.mcn_pack_x3_bytes.restype = [_i, _i], C.c_int64
    lib.mcn_pack_x3.argtypes, lib.mcn_pack_x3.restype = [C.POINTER(C.c_float), _i, _i, _vp], C.c_int
    lib.mcn_abi_version.restype = _i
    lib.mcn_sizeof.argtypes, lib.mcn_sizeof.restype = [_i], C.c_int64
    lib.mcn_last_dispatch.restype = C.c_char_p
    _check_abi(lib)
    return lib


ABI_VERSION = 5          # include/mcn.h: MCN_ABI_VERSION this binding's struct mirrors were written against


def _check_abi(lib):
    """A library built from another header must not receive these struct mirrors (include/mcn.h: ABI guard)."""
    got = int(lib.mcn_abi_version())
    if got != ABI_VERSION:
        raise ImportError("modelcrowdnav_amd: %s has ABI %d, this binding was written for ABI %d -- rebuild the library "
                          "(make -C modelcrowdnav_amd/csrc)" % (LIB_PATH, got, ABI_VERSION))
    mirrors = {0: EnvCfg, 1: EnvState, 2: EnvOut, 3: Rollout, 4: Tuning, 5: StepRec, 6: RollRec, 9: ScenarioCfg}
    for which, cls in mirrors.items():
        if int(lib.mcn_sizeof(which)) != C.sizeof(cls):
            raise ImportError("modelcrowdnav_amd: struct %s is %d bytes here, %d in %s" %
                              (cls.__name__, C.sizeof(cls), int(lib.mcn_sizeof(which)), LIB_PATH))


def last_dispatch():
    """Kernel family the calling thread's latest mcn_env_step / mcn_env_rollout dispatched to (mcn_last_dispatch)."""
    return lib.mcn_last_dispatch().decode()


lib = _load()

# every symbol include/mcn.h declares; tests/test_abi.py checks the .so exports each one
EXPORTED = ["mcn_version", "mcn_abi_version", "mcn_sizeof", "mcn_last_dispatch", "mcn_pack_x3", "mcn_pack_x3_bytes", "mcn_set_tuning", "mcn_get_tuning", "mcn_env_step", "mcn_env_lp3_queue_bytes", "mcn_env_rollout", "mcn_scenario_pool", "mcn_orca_batch", "mcn_pack_linear", "mcn_sarl_workspace_bytes",
            "mcn_sarl_lookahead", "mcn_sarl_lookahead_env", "mcn_sarl_predict", "mcn_sgan_workspace_bytes", "mcn_sgan_step", "mcn_mlp_world_step", "mcn_attn_world_workspace_bytes",
            "mcn_attn_world_step"]


def check(rc, what):
    if rc != MCN_OK:
        raise McnError("%s failed with code %d (%s)" % (what, rc, {-1: "MCN_EINVAL", -2: "MCN_ELAUNCH"}.get(rc, "?")))


def get_tuning():
    t = Tuning()
    check(lib.mcn_get_tuning(C.byref(t)), "mcn_get_tuning")
    return t


def set_tuning(**kw):
    """Override kernel dispatch (tests, tuning): set_tuning(quad_max_envs=0, force_generic=1); no arguments =
    back to the initial (automatic / environment) values.  The given fields are laid over the CURRENT settings.
    Returns the previous settings.  Process-wide; launches snapshot the settings under the library's lock."""
    prev = get_tuning()
    if not kw:
        check(lib.mcn_set_tuning(None), "mcn_set_tuning")
        return prev
    t = Tuning.from_buffer_copy(prev)            # overlay: a nested tuned(...) keeps the outer overrides
    for k, v in kw.items():
        if k not in dict(Tuning._fields_):
            raise TypeError("unknown tuning field %r" % k)
        setattr(t, k, int(v))
    check(lib.mcn_set_tuning(C.byref(t)), "mcn_set_tuning")
    return prev


class tuned:
    """Context manager: `with tuned(rollout_fused=1): ...` restores the previous dispatch on exit."""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.prev = set_tuning(**self.kw)
        return self

    def __exit__(self, *exc):
        check(lib.mcn_set_tuning(C.byref(self.prev)), "mcn_set_tuning")
        return False


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def version():
    return lib.mcn_version().decode()


def device_arch(device=None):
    """gcnArchName of the device the kernels will run on (queried through torch's runtime)."""
    return torch.cuda.get_device_properties(device if device is not None else torch.cuda.current_device()).gcnArchName
