"""Reference names of crowd_nav/utils/misc.py for the model-based drivers (train_model_based_sgan.py:29 does
`from crowd_nav.utils.misc import *`): GetRealData :47-116, StoreAction :119-124, PositiveRate :39-44.  The parsing,
scene joining, padding and SGAN cache writing live in utils/realdata.py (pinned by tests/golden/g9_realdata.npz);
this module only puts its arrays back into the containers the reference's callers expect."""
import logging
import os

import torch

from . import realdata


class RawMemory(object):
    """The reference's list-backed ReplayMemory (memory.py:4-34) for rows that are not (state, value) tensor pairs:
    GetRealData's `(ob, reward, done, info, start_ends)` rows and StoreAction's `(current_s, next_action)` pairs."""

    def __init__(self, capacity):
        self.capacity = int(capacity)
        self.memory = list()
        self.position = 0

    def push(self, item):
        if len(self.memory) < self.position + 1:
            self.memory.append(item)
        else:
            self.memory[self.position] = item
        self.position = (self.position + 1) % self.capacity

    def is_full(self):
        return len(self.memory) == self.capacity

    def __getitem__(self, item):
        return self.memory[item]

    def __len__(self):
        return len(self.memory)

    def clear(self):
        self.memory = list()


def PositiveRate(memory):
    """misc.py:39-44: share of stored values > 0.  Works on the device ring (utils/memory.py) and on list memories."""
    if hasattr(memory, "_values") and memory._values is not None:
        n = len(memory)
        return float((memory._values[:n] > 0).sum().item()) / n
    pos = sum(1 for _, value in memory.memory if value.item() > 0)
    return pos / len(memory.memory)


def StoreAction(memory, cur_obs, last_obs):
    """misc.py:119-124: one world-model training pair."""
    current_s = [o.getvalue() for o in last_obs]
    next_s = [o.getvalue() for o in cur_obs]
    next_action = [s[2:] for s in next_s][:len(current_s)]
    memory.push((torch.Tensor(current_s), torch.Tensor(next_action)))


def GetRealData(dataset_file, phase="train", capacity=10000, stride=-1, windows_size=-1, padding_last="stay",
                padding_first="none", dataset_slice=None, Store_for_world_fn=None, cacheFile=None):
    """misc.py:47-116: (raw_memory, rawob).  raw_memory rows are `(list[ObservableState], 0, done, Nothing(),
    start_ends)`; rawob is filled through `Store_for_world_fn(rawob, obs[i], obs[i - 1])` for every consecutive pair
    of frames of every scene; with `cacheFile` the SGAN text caches `<n>.txt` are written there."""
    from ..envs.utils.info import Nothing
    from ..envs.utils.state import ObservableState
    data = realdata.get_real_data(dataset_file, phase=phase, stride=stride, windows_size=windows_size,
                                  padding_last=padding_last, padding_first=padding_first, dataset_slice=dataset_slice,
                                  cache_dir=cacheFile)
    raw_memory, rawob = RawMemory(capacity), RawMemory(capacity)
    for sc in data.scenes:
        T = sc["obs"].shape[0]
        se = sc["start_ends"].tolist()
        obs = [[ObservableState(*r) for r in sc["obs"][t][sc["present"][t]].tolist()] for t in range(T)]
        for i, ob in enumerate(obs):
            raw_memory.push((ob, 0, i == T - 1, Nothing(), se))
            if i > 0 and Store_for_world_fn is not None:
                Store_for_world_fn(rawob, obs[i], obs[i - 1])
    logging.info("Loaded %s cases in %s for phase: %s " % (len(data.scenes), os.path.basename(dataset_file), phase))
    return raw_memory, rawob
