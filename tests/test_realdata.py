"""Recorded-data formats against what the REAL reference produced on the same file:
tests/golden/g9_realdata.npz = crowd_nav/utils/misc.py:GetRealData (with the vendored trajnetplusplustools reader)
run on tests/golden/g9_scenes.ndjson by tests/golden_tools/gen_golden_nets.py:g9_realdata.  Everything is compared exactly:
observation values are copies / one multiply of file values, cache files are compared byte for byte."""
import os

import numpy as np
import pytest

from modelcrowdnav_amd.utils import realdata as RD

CASES = [("default_test", dict(phase="test")),
         ("default_train", dict(phase="train")),
         ("default_val", dict(phase="val")),
         ("win_moving", dict(phase="test", stride=2, windows_size=6, padding_last="moving", padding_first="stay")),
         ("win_slice", dict(phase="val", stride=3, windows_size=5, dataset_slice=[1, 6]))]


@pytest.mark.parametrize("name,kw", CASES)
def test_get_real_data_matches_reference(name, kw, golden_dir, tmp_path):
    g = np.load(os.path.join(golden_dir, "g9_realdata.npz"))
    data = RD.get_real_data(os.path.join(golden_dir, "g9_scenes.ndjson"), cache_dir=str(tmp_path), **kw)
    raw = data.raw_memory()
    assert [len(r[0]) for r in raw] == g[name + "_count"].tolist()
    assert [bool(r[2]) for r in raw] == g[name + "_done"].tolist()
    flat = np.concatenate([r[0] for r in raw])
    assert np.array_equal(flat, g[name + "_obs"])
    se = np.concatenate([np.asarray(r[4]).reshape(-1, 4) for r in raw])
    assert [len(r[4]) for r in raw] == g[name + "_se_count"].tolist() and np.array_equal(se, g[name + "_se"])
    pairs = data.world_pairs()
    assert [p[0].shape[0] for p in pairs] == g[name + "_pair_count"].tolist()
    assert [p[1].shape[0] for p in pairs] == g[name + "_pair_next_count"].tolist()
    assert np.array_equal(np.concatenate([p[0] for p in pairs]), g[name + "_pair_cur"])
    assert np.array_equal(np.concatenate([p[1] for p in pairs]), g[name + "_pair_next"])
    files = sorted(os.listdir(tmp_path), key=lambda x: int(x.split(".")[0]))
    assert len(files) == int(g[name + "_cache_files"])
    for f in files:
        assert open(os.path.join(tmp_path, f), "rb").read() == g[name + "_cache_" + f.split(".")[0]].tobytes(), f
    # the reference's layout with its value types round-trips too
    states = data.raw_memory(as_states=True)
    assert states[0][0][0].px == flat[0, 0] and states[0][0][0].radius == 0.3


def test_cache_round_trip_and_history(golden_dir, tmp_path):
    data = RD.get_real_data(os.path.join(golden_dir, "g9_scenes.ndjson"), phase="test", cache_dir=str(tmp_path))
    rows = RD.read_sgan_cache(os.path.join(tmp_path, "2.txt"))
    assert rows.shape[1] == 4 and rows[0, 0] == 80.0
    p2 = os.path.join(tmp_path, "copy.txt")
    RD.write_sgan_cache(p2, rows)
    assert open(p2).read() == open(os.path.join(tmp_path, "2.txt")).read()
    hist, peds = RD.history_from_cache(rows, obs_len=8)
    frames = np.unique(rows[:, 0])[-8:]
    assert hist.shape == (8, len(peds), 2)
    for j, pid in enumerate(peds):                      # present samples are the file values rounded to 1e-4
        for i, f in enumerate(frames):
            m = rows[(rows[:, 0] == f) & (rows[:, 1] == pid)]
            if len(m):
                assert np.array_equal(hist[i, j], np.around(m[0, 2:4], 4))
    # a pedestrian that left before the window keeps standing at its last position
    left = [j for j, pid in enumerate(peds) if rows[rows[:, 1] == pid][:, 0].max() < frames[-1]]
    for j in left:
        last = rows[rows[:, 1] == peds[j]]
        assert np.array_equal(hist[-1, j], np.around(last[np.argmax(last[:, 0]), 2:4], 4))


def test_episode_tensor_feeds_datagen(golden_dir):
    data = RD.get_real_data(os.path.join(golden_dir, "g9_scenes.ndjson"), phase="test", stride=2, windows_size=6,
                            padding_last="moving", padding_first="stay")
    obs, lengths, ids = data.episode_tensor()
    assert obs.ndim == 4 and obs.shape[3] == 5 and (lengths == 7).all() and len(ids) == obs.shape[0]
    with pytest.raises(ValueError):
        RD.RealData([dict(id=0, obs=np.zeros((3, 2, 5)), present=np.array([[1, 0], [1, 1], [1, 1]], bool))], [], []).episode_tensor()


def test_reference_named_GetRealData_fills_the_callers_containers(golden_dir, tmp_path):
    """crowd_nav/utils/misc.py:47-116 as the model-based driver calls it (train_model_based_sgan.py:249-269):
    `raw_memory, rawob = GetRealData(..., Store_for_world_fn=StoreAction, cacheFile=dir)`; same fixture as above."""
    import torch
    from modelcrowdnav_amd.utils.misc import GetRealData, StoreAction, PositiveRate, RawMemory
    g = np.load(os.path.join(golden_dir, "g9_realdata.npz"))
    raw, rawob = GetRealData(os.path.join(golden_dir, "g9_scenes.ndjson"), phase="test", Store_for_world_fn=StoreAction,
                             cacheFile=str(tmp_path))
    name = "default_test"
    assert [len(r[0]) for r in raw.memory] == g[name + "_count"].tolist()
    assert [bool(r[2]) for r in raw.memory] == g[name + "_done"].tolist()
    flat = np.array([[o.px, o.py, o.vx, o.vy, o.radius] for r in raw.memory for o in r[0]])
    assert np.array_equal(flat, g[name + "_obs"])
    assert [tuple(p[0].shape)[0] for p in rawob.memory] == g[name + "_pair_count"].tolist()
    assert np.array_equal(torch.cat([p[0] for p in rawob.memory]).numpy(), g[name + "_pair_cur"])
    assert np.array_equal(torch.cat([p[1] for p in rawob.memory]).numpy(), g[name + "_pair_next"])
    assert len(os.listdir(tmp_path)) == int(g[name + "_cache_files"])
    m = RawMemory(3)                           # the ring semantics of memory.py:13-19
    for v in (1.0, -1.0, 2.0, 3.0):
        m.push((None, torch.tensor([v])))
    assert len(m) == 3 and m.is_full() and [float(x[1]) for x in m.memory] == [3.0, -1.0, 2.0]
    assert PositiveRate(m) == 2 / 3
