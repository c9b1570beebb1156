"""Multi-GPU plumbing for the rollout path: one process per GPU, torch.distributed
(backend 'nccl' = RCCL over xGMI on ROCm; 'gloo' in CPU tests).

The path shards by environment: envs are independent, so there is NO per-step collective.
The only exchange is one all_gather of finished-episode records per rollout (a few hundred KB
at most: latency-bound on xGMI, hence a single fused buffer and a single collective).
Seeds / scenario cases are functions of the GLOBAL env id, so results do not depend on the
number of ranks.
"""
import os

import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def init_from_env(backend=None, force=False):
    """Initialise from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run sets them).
    A one-rank job needs no process group; `force` creates one anyway (the world-size-1 RCCL test)."""
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if (ws <= 1 and not force) or dist.is_initialized():
        return world()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group(backend, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    return world()


def shard(total, rank, world_size):
    """Contiguous [lo, hi) slice of `total` global env ids owned by `rank` (sizes differ by at most 1)."""
    base, extra = divmod(total, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_records(returns, infos, times, counts=None, extras=(), equal_shards=None):
    """One collective: all ranks contribute [n_local] episode records and receive the
    concatenation in rank order.  returns/times float64, infos uint8/int; all on the same device.
    `extras`: further per-record float64 columns that ride in the same collective (returned as res["extras"]).
    `equal_shards`: True when the caller knows every rank holds the same number of records (E_total divisible by
    the rank count: one all_gather_into_tensor, no size exchange), False when it knows they differ; None = ask
    (one extra all_gather of the sizes).  Records travel as float64 columns, not the float32 + u8 SURVEY 8(e)
    sketches: the gathered returns are compared bit for bit with a single-process run, and the message is
    latency-bound on xGMI either way (32 768 episodes x 5 columns = 1.3 MB)."""
    n = returns.numel()
    cols = [returns.reshape(-1).double(), infos.reshape(-1).double(), times.reshape(-1).double()]
    if counts is not None:
        cols.append(counts.reshape(-1).double())
    cols += [x.reshape(-1).double() for x in extras]
    packed = torch.stack(cols, 1).contiguous()
    rank, ws = world()
    home = packed.device
    grouped = dist.is_available() and dist.is_initialized()
    if grouped and dist.get_backend() == "gloo" and packed.is_cuda:
        packed = packed.cpu()          # rehearsals / tests on one GPU: the collective runs on host copies
    if not grouped:
        out = packed
    else:
        sizes = None
        if equal_shards is None or not equal_shards:
            mine = torch.tensor([n], dtype=torch.int64, device=packed.device)
            got = torch.empty(ws, dtype=torch.int64, device=packed.device)
            dist.all_gather_into_tensor(got, mine)
            sizes = [int(x) for x in got.tolist()]
        if sizes is None or len(set(sizes)) == 1:
            out = torch.empty(ws * n, packed.shape[1], dtype=packed.dtype, device=packed.device)
            dist.all_gather_into_tensor(out, packed)
        else:
            m = max(sizes)
            pad = torch.zeros(m, packed.shape[1], dtype=packed.dtype, device=packed.device)
            pad[:n] = packed
            buf = torch.empty(ws * m, packed.shape[1], dtype=packed.dtype, device=packed.device)
            dist.all_gather_into_tensor(buf, pad)
            out = torch.cat([buf[r * m:r * m + sizes[r]] for r in range(ws)], 0)
    out = out.to(home)
    res = {"return": out[:, 0], "info": out[:, 1].to(torch.uint8), "time": out[:, 2]}
    if counts is not None:
        res["count"] = out[:, 3].to(torch.int32)
    first = 4 if counts is not None else 3
    res["extras"] = [out[:, first + i] for i in range(len(extras))]
    return res
