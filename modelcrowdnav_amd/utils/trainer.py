"""Value-network trainer (reference: crowd_nav/utils/trainer.py:19-82): SGD(momentum 0.9) + MSE on
(state, value) pairs from the replay memory.

Data-parallel: when torch.distributed is initialised every rank trains on its own memory shard and the
gradients are averaged with ONE all-reduce of a single flat bucket per step (96 502 floats = 386 KB: latency-
bound on xGMI, so one message, not one per tensor).  Weights start identical (broadcast from rank 0 in
`sync_weights`) and stay identical because every rank applies the same averaged gradient.  Ranks hold different
amounts of experience (episode lengths differ per env shard), so the number of steps of an epoch is agreed on ONCE
per epoch (all-reduce MAX of the local batch counts) and a rank that runs out of rows wraps around its own
permutation: every rank issues the same number of collectives, every batch is full.

The arithmetic (one SGD-momentum step of the MSE loss) is pinned against the reference's own Trainer by
tests/golden/g10_trainer.npz (tests/test_training_cpu.py).
"""
import logging

import torch
import torch.distributed as dist
import torch.nn as nn
import torch.optim as optim


class Trainer(object):
    def __init__(self, model, memory, device, batch_size):
        self.model = model
        self.device = device
        self.criterion = nn.MSELoss().to(device)
        self.memory = memory
        self.data_loader = None          # kept for attribute compatibility; batches come from memory.sample
        self.batch_size = batch_size
        self.optimizer = None
        self._flat = None
        self._gen = torch.Generator()

    def set_learning_rate(self, learning_rate):
        logging.info("Current learning rate: %f", learning_rate)
        self.optimizer = optim.SGD(self.model.parameters(), lr=learning_rate, momentum=0.9)

    # ------------------------------------------------------------------ data parallel helpers
    @staticmethod
    def _grouped():
        return dist.is_available() and dist.is_initialized()

    @staticmethod
    def _world():
        return dist.get_world_size() if Trainer._grouped() else 1

    @staticmethod
    def _collective(fn, t):
        """Run collective `fn` on tensor t where the backend can see it: RCCL ('nccl') works on device memory; under
        gloo (CPU rehearsals, several ranks sharing one GPU in tests) device tensors make the trip through a host copy."""
        if t.is_cuda and dist.get_backend() == "gloo":
            h = t.cpu()
            fn(h)
            t.copy_(h)
        else:
            fn(t)

    def sync_weights(self, src=0):
        if self._world() > 1:
            for p in self.model.parameters():
                self._collective(lambda t: dist.broadcast(t, src), p.data)

    def _allreduce_grads(self):
        # a process group of ONE rank still runs the collective (the world-size-1 RCCL test exercises exactly the
        # call an N-rank job makes); without a group there is nothing to reduce
        if not self._grouped():
            return
        ws = self._world()
        params = [p for p in self.model.parameters() if p.grad is not None]
        n = sum(p.grad.numel() for p in params)
        if self._flat is None or self._flat.numel() != n or self._flat.device != params[0].grad.device:
            self._flat = torch.empty(n, dtype=params[0].grad.dtype, device=params[0].grad.device)
        off = 0
        for p in params:
            k = p.grad.numel()
            self._flat[off:off + k].copy_(p.grad.reshape(-1))
            off += k
        self._collective(dist.all_reduce, self._flat)          # one bucket, one collective
        if ws > 1:
            self._flat.div_(ws)
        off = 0
        for p in params:
            k = p.grad.numel()
            p.grad.copy_(self._flat[off:off + k].view_as(p.grad))
            off += k

    def _step(self, inputs, values):
        self.optimizer.zero_grad()
        loss = self.criterion(self.model(inputs), values)
        loss.backward()
        self._allreduce_grads()
        self.optimizer.step()
        return loss.data.item()

    # ------------------------------------------------------------------ reference surface
    def _agreed_steps(self, n_local):
        """Mini-batches per epoch: the local count, or with several ranks the largest count of any rank."""
        steps = -(-n_local // self.batch_size)
        if self._world() > 1:
            dev = self.device if dist.get_backend() == "nccl" else torch.device("cpu")
            t = torch.tensor([steps], dtype=torch.int64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            steps = int(t.item())
        return steps

    def optimize_epoch(self, num_epochs, perms=None):
        """trainer.py:38-62: full passes over the memory in shuffled mini-batches; returns epoch_loss / len(memory).
        perms (optional, [num_epochs][n] index rows) replaces the shuffles this trainer would draw itself -- the
        reference's DataLoader draws them from torch's global generator."""
        if self.optimizer is None:
            raise ValueError("Learning rate is not set!")
        average_epoch_loss = 0
        n = len(self.memory)
        multi = self._world() > 1
        for ep in range(num_epochs):
            epoch_loss = 0
            perm = torch.randperm(n, generator=self._gen) if perms is None else torch.as_tensor(perms[ep], dtype=torch.long)
            steps = self._agreed_steps(n)
            for b in range(steps):
                lo = b * self.batch_size
                if multi:
                    # every rank runs `steps` full batches; a rank with fewer rows wraps around its permutation
                    idx = perm[(lo + torch.arange(self.batch_size)) % n]
                else:
                    idx = perm[lo:lo + self.batch_size]
                idx = idx.to(self.memory._states.device)
                epoch_loss += self._step(self.memory._states[idx].to(self.device), self.memory._values[idx].to(self.device))
            average_epoch_loss = epoch_loss / n
        return average_epoch_loss

    def optimize_batch(self, num_batches):
        """trainer.py:64-82: num_batches independent random mini-batches; returns the mean loss."""
        if self.optimizer is None:
            raise ValueError("Learning rate is not set!")
        losses = 0
        for _ in range(num_batches):
            inputs, values = self.memory.sample(self.batch_size, self._gen)
            losses += self._step(inputs.to(self.device), values.to(self.device))
        average_loss = losses / num_batches
        logging.debug("Average loss : %.2E", average_loss)
        return average_loss
