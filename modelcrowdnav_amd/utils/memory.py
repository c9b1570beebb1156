"""Replay memory (reference: crowd_nav/utils/memory.py:4-34).

Same Dataset surface (push / is_full / __getitem__ / __len__ / clear / shuffle / capacity / position), but the
experience lives in two device tensors -- states [capacity, N, 13] float32 and values [capacity, 1] float32 --
so a whole batched rollout is appended with one copy (`push_batch`) and a training batch is one gather
(`sample`).  Items of a different human count than the buffer was created with are kept in a small host-side
overflow list, which is what the reference's collate_fn effectively drops (trainer.py:9-17).
"""
import random

import torch
from torch.utils.data import Dataset


class ReplayMemory(Dataset):
    def __init__(self, capacity, init_value=None, device=None):
        self.capacity = int(capacity)
        self.position = 0
        self._len = 0
        self._states = None
        self._values = None
        self._device = device
        self._odd = {}            # index -> item whose state shape differs from the tensor store
        if init_value is not None:
            for _ in range(self.capacity):
                self.push(init_value)

    # ------------------------------------------------------------------ storage
    def _ensure(self, state):
        if self._states is None:
            dev = self._device if self._device is not None else state.device
            self._states = torch.zeros((self.capacity,) + tuple(state.shape), dtype=torch.float32, device=dev)
            self._values = torch.zeros(self.capacity, 1, dtype=torch.float32, device=dev)

    def push(self, item):
        """Append one (state [N,13], value [1]) pair, overwriting the oldest when full (memory.py:13-19)."""
        state, value = item
        self._ensure(state)
        pos = self.position
        if tuple(state.shape) == tuple(self._states.shape[1:]):
            self._states[pos].copy_(state)
            self._values[pos].copy_(value.reshape(1))
            self._odd.pop(pos, None)
        else:
            self._odd[pos] = (state, value)
        self._len = max(self._len, pos + 1)
        self.position = (pos + 1) % self.capacity

    def push_batch(self, states, values):
        """states [B,N,13], values [B] or [B,1]: appended in order with ring wrap-around."""
        B = states.shape[0]
        if B == 0:
            return
        self._ensure(states[0])
        values = values.reshape(B, 1).to(self._values.dtype)
        if B >= self.capacity:                       # only the newest `capacity` survive
            states, values, B = states[-self.capacity:], values[-self.capacity:], self.capacity
        idx = (self.position + torch.arange(B, device=self._states.device)) % self.capacity
        self._states.index_copy_(0, idx, states.to(self._states.device, torch.float32))
        self._values.index_copy_(0, idx, values.to(self._values.device))
        for k in idx.tolist():
            self._odd.pop(k, None)
        self._len = min(self.capacity, max(self._len, self.position + B))
        self.position = (self.position + B) % self.capacity

    def sample(self, batch_size, generator=None):
        """Uniform batch without replacement within the batch -> (states [b,N,13], values [b,1])."""
        n = len(self)
        b = min(batch_size, n)
        perm = torch.randperm(n, generator=generator)[:b].to(self._states.device)
        return self._states[perm], self._values[perm]

    # ------------------------------------------------------------------ Dataset surface
    def is_full(self):
        return self._len == self.capacity

    def __getitem__(self, item):
        if item in self._odd:
            return self._odd[item]
        if not 0 <= item < self._len:
            raise IndexError(item)
        return self._states[item], self._values[item]

    def __len__(self):
        return self._len

    def clear(self):
        self._len, self.position, self._odd = 0, 0, {}

    def shuffle(self):
        n = self._len
        if n:
            perm = list(range(n))
            random.shuffle(perm)
            p = torch.tensor(perm, device=self._states.device)
            self._states[:n] = self._states[p]
            self._values[:n] = self._values[p]

    @property
    def memory(self):
        """List view, for callers that read `.memory` directly (train.py:219)."""
        return [self[i] for i in range(self._len)]
