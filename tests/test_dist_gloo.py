"""CPU, world_size 2, gloo: the N > 1 plumbing of the rollout path (sharding by global env id and the
single all_gather of episode records).  No GPU compute is involved: records are synthetic."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, ws, port, E_total, rounds, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(ws), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from modelcrowdnav_amd import dist as mdist
    r, w = mdist.init_from_env("gloo")
    assert (r, w) == (rank, ws)
    lo, hi = mdist.shard(E_total, rank, ws)
    gid = torch.arange(lo, hi, dtype=torch.float64)
    # record of episode (global env g, round k) is a pure function of (g, k): partition-invariant by construction
    ret = torch.stack([gid * 10 + k for k in range(rounds)], 1)
    info = torch.stack([((gid.long() + k) % 3 + 2).to(torch.uint8) for k in range(rounds)], 1)
    tim = ret / 4
    rec = mdist.gather_records(ret, info, tim, equal_shards=(E_total % ws == 0))
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), ret=rec["return"].numpy(), info=rec["info"].numpy(),
             time=rec["time"].numpy())
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_shard_covers_everything():
    from modelcrowdnav_amd import dist as mdist
    for total in (1, 7, 8, 4096, 32768):
        for ws in (1, 2, 3, 8):
            spans = [mdist.shard(total, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


@pytest.mark.parametrize("E_total", [64, 33])          # 33 on two ranks: shards of 17 and 16 records x rounds
def test_gather_records_world2(tmp_path, E_total):
    rounds, ws = 3, 2
    port = 29500 + (os.getpid() % 2000) + E_total % 7
    mp.spawn(_worker, args=(ws, port, E_total, rounds, str(tmp_path)), nprocs=ws, join=True)
    outs = [np.load(os.path.join(str(tmp_path), "r%d.npz" % r)) for r in range(ws)]
    g = np.arange(E_total, dtype=np.float64)
    want_ret = np.stack([g * 10 + k for k in range(rounds)], 1).reshape(-1)
    want_info = np.stack([(g.astype(np.int64) + k) % 3 + 2 for k in range(rounds)], 1).reshape(-1)
    for o in outs:                       # every rank receives the same, rank-ordered concatenation
        assert np.array_equal(o["ret"], want_ret)
        assert np.array_equal(o["info"], want_info)
        assert np.array_equal(o["time"], want_ret / 4)
