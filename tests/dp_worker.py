"""Child process of tests/test_dist_gpu.py: one rank of a data-parallel value-network Trainer ON THE DEVICE
(crowd_nav/utils/trainer.py:64-82 + the one flat-bucket gradient all-reduce per step of utils/trainer.py).  Started
as a fresh `python -m tests.dp_worker <rank> <world> <port> <backend> <mode> <out.pt>` process; every rank uses GPU 0
(the GPU box has one card: `gloo` carries the collectives on host copies for world 2, `nccl` = RCCL for world 1).

modes
  g10      the reference Trainer's own run (tests/golden/g10_trainer.npz, `batch`: three optimize_batch(1) calls over a
           one-batch memory, lr 0.01) split over the ranks: rank r holds rows r::world of the memory and its batch is its
           whole shard, so the averaged gradient is the gradient of the reference's batch.
  unequal  ranks hold 32 vs 45 rows (2 vs 3 mini-batches of 16): epochs must stay in lock step.
"""
import os
import sys


def main():
    rank, ws, port, backend, mode, out = sys.argv[1:7]
    rank, ws = int(rank), int(ws)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(ws), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist
    from modelcrowdnav_amd import dist as mdist
    from modelcrowdnav_amd.utils.memory import ReplayMemory
    from modelcrowdnav_amd.utils.trainer import Trainer
    from tests.test_training_cpu import _load_sd, _model
    mdist.init_from_env(backend, force=True)
    dev = torch.device("cuda", 0)
    torch.set_num_threads(1)
    if mode == "g10":
        g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g10_trainer.npz"))
        model = _model(seed=3 + rank)                     # deliberately different per rank: sync_weights must fix it
        if rank == 0:
            _load_sd(model, g, "w0__")
        model.to(dev)
        states, values = torch.from_numpy(g["batch_states"]), torch.from_numpy(g["batch_values"]).reshape(-1)
        mine = slice(rank, None, ws)
        mem = ReplayMemory(states[mine].shape[0], device=dev)
        mem.push_batch(states[mine].to(dev), values[mine].to(dev))
        tr = Trainer(model, mem, dev, len(mem))
        tr.sync_weights()
        tr.set_learning_rate(0.01)
        losses = [tr.optimize_batch(1) for _ in range(3)]
    else:
        model = _model(seed=0).to(dev)
        gen = torch.Generator().manual_seed(11)
        states = torch.randn(77, 5, 13, generator=gen)
        values = torch.randn(77, generator=gen)
        rows = slice(0, 32) if rank == 0 else slice(32, 77)
        mem = ReplayMemory(64, device=dev)
        mem.push_batch(states[rows].to(dev), values[rows].to(dev))
        tr = Trainer(model, mem, dev, 16)
        tr.sync_weights()
        tr.set_learning_rate(0.02)
        losses = [tr.optimize_epoch(3), tr.optimize_batch(2)]      # would hang if the ranks disagreed on the step count
    assert all(p.is_cuda for p in model.parameters()) and tr._flat is not None and tr._flat.is_cuda
    torch.save({"weights": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, "losses": losses,
                "backend": dist.get_backend(), "world": dist.get_world_size(), "rows": len(mem)}, out)
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
