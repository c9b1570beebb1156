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
    assert ctypes.sizeof(_hip.EnvState) == 14 * 8 and _hip.EnvState.hcount.offset == 13 * 8
    assert ctypes.sizeof(_hip.EnvOut) == 5 * 8 and _hip.EnvOut.lp3_queue.offset == 32
    assert ctypes.sizeof(_hip.Tuning) == 10 * 4
    assert ctypes.sizeof(_hip.StepRec) == 24 and _hip.StepRec.done.offset == 16 and _hip.StepRec.hh_count.offset == 20
    assert ctypes.sizeof(_hip.RollRec) == 32 and _hip.RollRec.fin_count.offset == 12
    assert _hip.RollRec.danger_dist_sum.offset == 24
    r = _hip.Rollout
    assert r.disc_len.offset == 8 and r.danger_episodes.offset == 12 and r.state.offset == 16
    assert r.fin_slots.offset == 48 and r.danger_short_from.offset == 52
    assert r.pool_hpos.offset == 56 and r.pool_size.offset == 96 and r.case_stride.offset == 100
    assert r.robot_start.offset == 104 and ctypes.sizeof(r) == 144


def test_record_views_alias_the_packed_records():
    """The named per-env tensors are strided views of mcn_step_rec / mcn_roll_rec arrays."""
    import numpy as np
    import torch
    from modelcrowdnav_amd import _hip
    raw = np.zeros(5, dtype=np.dtype([("reward", "f8"), ("dmin", "f8"), ("done", "u1"), ("info", "u1"),
                                      ("pad", "u2"), ("hh", "i4")]))
    raw["reward"], raw["dmin"], raw["done"], raw["info"], raw["hh"] = np.arange(5), -np.arange(5), 1, [0, 1, 2, 3, 4], 7
    v = _hip.step_rec_views(torch.from_numpy(raw.view(np.float64).reshape(5, 3).copy()))
    assert v["reward"].tolist() == [0, 1, 2, 3, 4] and v["dmin"][2] == -2 and v["done"].tolist() == [1] * 5
    assert v["info"].tolist() == [0, 1, 2, 3, 4] and v["hh_count"].tolist() == [7] * 5
    rr = np.zeros(3, dtype=np.dtype([("ret", "f8"), ("steps", "i4"), ("fin", "i4"), ("case", "i4"), ("dc", "i4"),
                                     ("dsum", "f8")]))
    rr["ret"], rr["steps"], rr["fin"], rr["case"], rr["dc"], rr["dsum"] = 0.5, 3, 4, [9, 8, 7], 2, 1.25
    w = _hip.roll_rec_views(torch.from_numpy(rr.view(np.float64).reshape(3, 4).copy()))
    assert w["ep_return"].tolist() == [0.5] * 3 and w["ep_steps"].tolist() == [3] * 3 and w["fin_count"][1] == 4
    assert w["next_case"].tolist() == [9, 8, 7] and w["danger_count"][0] == 2 and w["danger_dist_sum"][2] == 1.25


def test_bad_arguments_are_rejected_on_host():
    """Validation happens before any launch, so it is safe to exercise without a GPU."""
    from modelcrowdnav_amd import _hip
    cfg, st, out = _hip.EnvCfg(), _hip.EnvState(), _hip.EnvOut()
    rc = _hip.lib.mcn_env_step(cfg, st, None, None, out, None, 4, 5, 1, None)
    assert rc == _hip.MCN_EINVAL
    rc = _hip.lib.mcn_orca_batch(None, None, None, None, 1, 1, 10.0, 10, 5.0, 0.25, None)
    assert rc == _hip.MCN_EINVAL


def test_more_bad_arguments_are_rejected_on_host():
    """mcn_env_rollout / mcn_scenario_pool / mcn_sgan_step validation (host side, no launch)."""
    import ctypes as C
    from modelcrowdnav_amd import _hip
    cfg, st, out = _hip.EnvCfg(), _hip.EnvState(), _hip.EnvOut()
    fake = C.c_void_p(0x1000)                     # never dereferenced: validation fails first
    assert _hip.lib.mcn_env_rollout(cfg, st, fake, 0, out, None, 4, 5, None) == _hip.MCN_EINVAL      # T <= 0
    assert _hip.lib.mcn_env_rollout(cfg, st, None, 8, out, None, 4, 5, None) == _hip.MCN_EINVAL      # no actions
    assert _hip.lib.mcn_env_rollout(cfg, st, fake, 8, out, None, 4, 5, None) == _hip.MCN_EINVAL      # empty state
    sc = _hip.ScenarioCfg(4.0, 10.0, 0.2, 0.3, 1.0, 0.3, (0.0, -4.0), (0.0, 4.0), _hip.RULE_CIRCLE, 0)
    assert _hip.lib.mcn_scenario_pool(sc, 1, 0, 8, 5, None, fake, fake, fake, None) == _hip.MCN_EINVAL
    assert _hip.lib.mcn_scenario_pool(sc, 1, 0, 0, 5, fake, fake, fake, fake, None) == _hip.MCN_EINVAL   # P <= 0
    assert _hip.lib.mcn_scenario_pool(sc, 1, 0, 8, 99, fake, fake, fake, fake, None) == _hip.MCN_EINVAL  # N too large
    bad = _hip.ScenarioCfg(4.0, 10.0, 0.2, 0.3, 1.0, 0.3, (0.0, -4.0), (0.0, 4.0), 7, 0)
    assert _hip.lib.mcn_scenario_pool(bad, 1, 0, 8, 5, fake, fake, fake, fake, None) == _hip.MCN_EINVAL  # unknown rule
    assert _hip.lib.mcn_sgan_step(None, fake, 0, 1, None, fake, None, fake, fake, None, 0.25, 4, 5, None) == _hip.MCN_EINVAL
    r = _hip.Rollout()
    assert ctypes.sizeof(_hip.ScenarioCfg) == 10 * 8 + 2 * 4


def test_abi_version_and_struct_sizes_are_checked_against_the_library():
    """include/mcn.h's ABI guard: the library reports the ABI it was built with and the size of every struct of the
    header; the binding refuses a library of another ABI at import (no GPU needed for either call)."""
    import pytest
    from modelcrowdnav_amd import _hip
    hdr = open(os.path.join(ROOT, "include", "mcn.h")).read()
    want = int(re.search(r"#define\s+MCN_ABI_VERSION\s+(\d+)", hdr).group(1))
    assert _hip.lib.mcn_abi_version() == want == _hip.ABI_VERSION
    for which, cls in {0: _hip.EnvCfg, 1: _hip.EnvState, 2: _hip.EnvOut, 3: _hip.Rollout, 4: _hip.Tuning, 5: _hip.StepRec,
                       6: _hip.RollRec, 9: _hip.ScenarioCfg}.items():
        assert _hip.lib.mcn_sizeof(which) == ctypes.sizeof(cls), cls.__name__
    assert _hip.lib.mcn_sizeof(7) > 0 and _hip.lib.mcn_sizeof(8) > 0 and _hip.lib.mcn_sizeof(99) == -1

    class Old(object):                                   # a library built from an older header
        def mcn_abi_version(self):
            return want - 1
    with pytest.raises(ImportError):
        _hip._check_abi(Old())
    assert _hip.last_dispatch() == ""                    # nothing launched in this process yet


def test_pack_x3_splits_every_weight_into_three_bfloat16_pieces():
    """mcn_pack_x3 (host helper, include/mcn.h): float32 fragments [NT][KT][64][4] -> [NT][ceil(KT / 2)][3][64][8]
    bfloat16; hi + mid + lo reproduces the float32 weight exactly (3 x 8 significand bits cover its 24), slot 8 q + s
    of input block m is slot 4 q + (s & 3) of k-tile 2 m + (s >> 2), an odd last k-tile is zero-padded."""
    import numpy as np
    from modelcrowdnav_amd import _hip
    rng = np.random.RandomState(0)
    for NT, KT in ((3, 7), (2, 10), (1, 1), (4, 5)):
        wf = (rng.normal(0, 1, (NT, KT, 64, 4)) * np.exp(rng.uniform(-20, 5, (NT, KT, 64, 4)))).astype(np.float32)
        wf[0, 0, :4, 0] = [0.0, -0.0, 1.0, -3.5]
        KB = (KT + 1) // 2
        assert _hip.lib.mcn_pack_x3_bytes(NT, KT) == NT * KB * 3 * 64 * 8 * 2
        out = np.zeros((NT, KB, 3, 64, 8), np.uint16)
        assert _hip.lib.mcn_pack_x3(wf.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), NT, KT, out.ctypes.data) == 0
        pieces = (out.astype(np.uint32) << 16).view(np.float32)                  # bfloat16 -> float32, exact
        total = pieces[:, :, 0].astype(np.float64) + pieces[:, :, 1] + pieces[:, :, 2]
        want = np.zeros((NT, KB, 64, 8), np.float64)
        for m in range(KB):
            for half in range(2):
                if 2 * m + half < KT:
                    want[:, m, :, 4 * half:4 * half + 4] = wf[:, 2 * m + half]
        assert np.array_equal(total, want)                                        # exact: no bit of the weight is lost
        assert np.all(np.abs(pieces[:, :, 1]) <= np.abs(pieces[:, :, 0]) * 2.0 ** -7 + 1e-45)
