"""World-model trainer (reference: crowd_nav/utils/trainer_sim.py:26-110 + pytorchtools.py EarlyStopping): Adam + MSE
on (humans' [px,py,vx,vy], their next velocities) pairs -- the `rawob` rows the explorers collect
(explorer.py:84-86) or `RealData.world_pairs()` -- with an 80 / 20 train / validation split, early stopping
(patience 7) on the validation loss, the best weights restored at the end and `model.mse` set to the best validation
loss (MlpWorld.noise_pre uses it).

The pairs are stacked into two device tensors once and mini-batches are index slices of them (the reference builds
two torch DataLoaders over Python lists per call); pairs whose pedestrian count differs from the first pair's are
dropped, as its collate_fn does (trainer_sim.py:15-23).
"""
import copy
import logging
import random

import torch
import torch.nn as nn
import torch.optim as optim


class EarlyStopping(object):
    """pytorchtools.py:6-52: stop after `patience` epochs without a validation-loss improvement of more than delta;
    the best weights are kept (in memory, and at `path` when given)."""

    def __init__(self, patience=7, delta=0, path=None):
        self.patience, self.delta, self.path = patience, delta, path
        self.counter, self.best_score, self.early_stop = 0, None, False
        self.best_state = None

    def __call__(self, val_loss, model):
        score = -val_loss
        if self.best_score is None or not (score < self.best_score + self.delta):
            self.best_score = score
            self.best_state = copy.deepcopy(model.state_dict())
            if self.path is not None:
                torch.save(self.best_state, self.path)
            self.counter = 0
        else:
            self.counter += 1
            if self.counter >= self.patience:
                self.early_stop = True


class Trainer_Sim(object):
    def __init__(self, model, memory, device, batch_size, path=None):
        self.model, self.memory, self.device, self.batch_size = model, memory, device, batch_size
        self.criterion = nn.MSELoss().to(device)
        self.optimizer = None
        self.train_size = 0.8
        self.patience = 7
        self.path = path
        self.early_stopping = EarlyStopping(patience=self.patience, path=path)

    def set_learning_rate(self, learning_rate):
        logging.info("Current learning rate: %f", learning_rate)
        self.optimizer = optim.Adam(self.model.parameters(), lr=learning_rate)

    def _pairs(self):
        rows = self.memory.memory if hasattr(self.memory, "memory") else self.memory
        if not isinstance(rows, list):
            rows = list(rows)
        random.shuffle(rows)            # trainer_sim.py:55: Python's RNG, and IN PLACE -- the memory itself is reordered,
        #                                 so a second call splits what the first one left
        n_train = int(len(rows) * self.train_size)

        def stack(part):
            if not part:
                return None, None
            shape = tuple(part[0][0].shape)
            part = [p for p in part if tuple(p[0].shape) == shape]
            cur = torch.stack([torch.as_tensor(p[0], dtype=torch.float32) for p in part]).to(self.device)
            nxt = torch.stack([torch.as_tensor(p[1], dtype=torch.float32) for p in part]).to(self.device)
            return cur.reshape(cur.shape[0], -1), nxt.reshape(nxt.shape[0], -1)
        return stack(rows[:n_train]), stack(rows[n_train:])

    def optimize_epoch(self, num_epochs, reset=False, perms=None):
        """trainer_sim.py:48-110.  Returns the best validation loss.
        perms (optional): (train [num_epochs][n_train], validation [num_epochs][n_val]) row orders that replace the
        shuffles drawn here -- the reference's two DataLoaders draw theirs from torch's global generator."""
        if self.optimizer is None:
            raise ValueError("Learning rate is not set!")
        (tx, ty), (vx, vy) = self._pairs()
        if tx is None or vx is None:
            raise ValueError("not enough pairs to split into training and validation sets")
        es = self.early_stopping
        es.counter, es.early_stop = 0, False
        if reset:
            es.best_score = None
        B = self.batch_size
        for ep in range(num_epochs):
            self.model.train()
            if perms is None:
                perm = torch.randperm(tx.shape[0], device=tx.device)
            else:
                perm = torch.as_tensor(perms[0][ep], dtype=torch.long, device=tx.device)
                vperm = torch.as_tensor(perms[1][ep], dtype=torch.long, device=vx.device)
            for i in range(0, tx.shape[0], B):
                idx = perm[i:i + B]
                loss = self.criterion(self.model(tx[idx]), ty[idx])
                self.optimizer.zero_grad()
                loss.backward()
                self.optimizer.step()
            self.model.eval()
            with torch.no_grad():
                ex, ey = (vx, vy) if perms is None else (vx[vperm], vy[vperm])
                losses = [self.criterion(self.model(ex[i:i + B]), ey[i:i + B]).item() for i in range(0, ex.shape[0], B)]
            es(sum(losses) / len(losses), self.model)
            if es.early_stop:
                break
        self.model.load_state_dict(es.best_state)               # load best model
        self.model.mse = 0 - es.best_score                      # used to add noise to the MLP world
        return -es.best_score
