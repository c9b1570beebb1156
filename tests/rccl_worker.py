"""Child process of tests/test_dist_gpu.py: a world-size-1 `nccl` (= RCCL on ROCm) process group on GPU 0.
Started as a fresh `python -m tests.rccl_worker <port> <out.json>` process, never forked or re-exec'd from pytest.
"""
import json
import os
import sys


def main():
    port, out = sys.argv[1:3]
    os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from modelcrowdnav_amd import dist as mdist
    from modelcrowdnav_amd.rollout import VecExplorer
    from tests import helpers as H
    from tests.test_rollout_gpu import _goal_seeking
    rank, ws = mdist.init_from_env("nccl", force=True)
    res = {"backend": dist.get_backend(), "world": ws}
    dev = torch.device("cuda", 0)
    g = torch.arange(48, dtype=torch.float64, device=dev)
    ret, info, tim = g * 10 + 0.125, (g.long() % 3 + 2).to(torch.uint8), g / 4
    a = mdist.gather_records(ret, info, tim, equal_shards=True)          # one all_gather_into_tensor
    b = mdist.gather_records(ret, info, tim, equal_shards=False)         # size exchange + padded gather
    ok = lambda r: bool(torch.equal(r["return"], ret) and torch.equal(r["info"], info) and torch.equal(r["time"], tim))
    res.update(gather_equal_ok=ok(a), gather_sizes_ok=ok(b), records_on_cuda=bool(a["return"].is_cuda))
    env = H.make_vec_env(32, 5)
    env.track_human_times = False
    env.export_human_actions = False
    ex = VecExplorer(env, env.robot, gamma=0.9, policy=object())
    r = ex.run_k_episodes(80, "test", action_fn=_goal_seeking, returnNav=True, total_envs=32)
    res.update(result=list(r), records=ex.last_records)
    json.dump(res, open(out, "w"))
    dist.barrier()
    torch.cuda.synchronize()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
