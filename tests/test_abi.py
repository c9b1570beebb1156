"""CPU: libmcn_hip.so loads without a GPU and exports every function include/mcn.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "mcn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mcn_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    from modelcrowdnav_amd import _hip
    names = declared_functions()
    assert names, "no declarations parsed"
    lib = ctypes.CDLL(_hip.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libmcn_hip.so does not export %s" % n
    assert sorted(_hip.EXPORTED) == names
    assert "gfx950" in _hip.version()


def test_struct_sizes_match_header():
    """ctypes mirrors must have the C layout (all members naturally aligned, no packing surprises)."""
    from modelcrowdnav_amd import _hip
    assert ctypes.sizeof(_hip.EnvCfg) == 7 * 8 + 2 * 4 + 6 * 4
    assert ctypes.sizeof(_hip.EnvState) == 13 * 8
    assert ctypes.sizeof(_hip.EnvOut) == 8 * 8
    r = _hip.Rollout
    assert r.disc_len.offset == 8 and r.ep_return.offset == 16 and r.fin_slots.offset == 64
    assert r.danger_count.offset == 72 and r.pool_hpos.offset == 88 and r.pool_size.offset == 128
    assert r.robot_start.offset % 8 == 0


def test_bad_arguments_are_rejected_on_host():
    """Validation happens before any launch, so it is safe to exercise without a GPU."""
    from modelcrowdnav_amd import _hip
    cfg, st, out = _hip.EnvCfg(), _hip.EnvState(), _hip.EnvOut()
    rc = _hip.lib.mcn_env_step(cfg, st, None, None, out, None, 4, 5, 1, None)
    assert rc == _hip.MCN_EINVAL
    rc = _hip.lib.mcn_orca_batch(None, None, None, None, 1, 1, 10.0, 10, 5.0, 0.25, None)
    assert rc == _hip.MCN_EINVAL
