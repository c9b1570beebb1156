"""Host logic of the batched DataGen (no GPU): episode indexing of raw_memory (datagen.py:220-230), the sample list
drawn in the reference's order (:236-241,412-418), rejection of ragged recordings."""
import random
import types

import numpy as np
import pytest
import torch


def _datagen(raw):
    from modelcrowdnav_amd.utils.datagen import VecDataGen
    env = types.SimpleNamespace(device=torch.device("cpu"))
    pol = types.SimpleNamespace(gamma=0.9)
    dg = VecDataGen(None, None, env, pol)
    dg.raw_memory = raw
    return dg


def _raw(lengths, N=3):
    raw = []
    for e, T in enumerate(lengths):
        for t in range(T):
            raw.append((np.full((N, 5), 10.0 * e + t), 0, t == T - 1, None))
    return raw


def test_episode_index_and_tensors():
    dg = _datagen(_raw([5, 12, 9]))
    assert dg.get_episode_start_index() == [0, 5, 17] and dg.count() == 3
    ep = dg.load_real_episodes()
    assert tuple(ep["obs"].shape) == (3, 12, 3, 5) and ep["length"] == [5, 12, 9]
    assert float(ep["obs"][1, 11, 0, 0]) == 21.0 and float(ep["obs"][0, 5, 0, 0]) == 0.0      # zero padded


def test_sample_list_follows_reference_draw_order():
    dg = _datagen(_raw([5, 12, 9, 30]))
    dg.load_real_episodes()
    # sequential picks: episodes in order, the 5-frame one skipped (len <= min_end), counter advances past it
    picks = dg._draw_samples(4, min_end=8, static_end=-1, add_sim=False, random_epi=False, test_case=None)
    assert [p[0] for p in picks] == [1, 2, 3, 1] and [p[1] for p in picks] == [12, 9, 30, 12] and dg.counter == 6
    # random picks: random.choice(indexes) then random.randrange(min_end, L), exactly the reference's two draws
    random.seed(5)
    picks = dg._draw_samples(6, min_end=8, static_end=-1, add_sim=True, random_epi=True, test_case=None)
    random.seed(5)
    want = []
    while len(want) < 6:
        i = random.choice([0, 5, 17, 26])
        L = {0: 5, 5: 12, 17: 9, 26: 30}[i]
        if L <= 8:
            continue
        want.append(({0: 0, 5: 1, 17: 2, 26: 3}[i], random.randrange(8, L)))
    assert picks == want
    # static_end overrides the drawn length but the draw still happens
    random.seed(5)
    picks = dg._draw_samples(3, min_end=8, static_end=9, add_sim=True, random_epi=True, test_case=None)
    assert [p[1] for p in picks] == [9, 9, 9] and [p[0] for p in picks] == [w[0] for w in want[:3]]


def test_growing_crowds_load_and_shrinking_ones_are_rejected():
    raw = _raw([6, 6])
    raw[2] = (np.zeros((4, 5)), 0, False, None)              # 3, 3, 4, 3, ... pedestrians: someone vanished
    with pytest.raises(NotImplementedError):
        _datagen(raw).load_real_episodes()
    grow = _raw([6, 6])
    for t in (3, 4, 5):
        grow[t] = (np.full((5, 5), 7.0), 0, t == 5, None)    # two pedestrians enter at frame 3 of episode 0
    ep = _datagen(grow).load_real_episodes()
    assert tuple(ep["obs"].shape) == (2, 6, 5, 5) and ep["count"][0].tolist() == [3, 3, 3, 5, 5, 5]
    assert ep["count"][1].tolist() == [3] * 6 and float(ep["obs"][0, 0, 4, 0]) == 0.0 and ep["first"][0].shape == (3, 5)
    with pytest.raises(RuntimeError):
        dg = _datagen(_raw([5, 6]))
        dg.load_real_episodes()
        dg._draw_samples(1, min_end=8, static_end=-1, add_sim=True, random_epi=True, test_case=None)
