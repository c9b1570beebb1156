"""Child process of tests/test_dist_gpu.py: one rank of a sharded VecExplorer rollout (gloo rendezvous on 127.0.0.1,
all ranks share GPU 0).  Started as a fresh `python -m tests.dist_worker` process, never forked from pytest.

    python -m tests.dist_worker <rank> <world> <port> <E_total> <N> <k> <out.json>
"""
import json
import os
import sys


def main():
    rank, ws, port, E_total, N, k, out = sys.argv[1:8]
    rank, ws, E_total, N, k = int(rank), int(ws), int(E_total), int(N), int(k)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(ws), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    torch.cuda.set_device(0)
    from modelcrowdnav_amd import dist as mdist
    from modelcrowdnav_amd.rollout import VecExplorer
    from tests import helpers as H
    from tests.test_rollout_gpu import _goal_seeking
    mdist.init_from_env("gloo")
    lo, hi = mdist.shard(E_total, rank, ws)
    env = H.make_vec_env(hi - lo, N)
    env.track_human_times = False
    env.export_human_actions = False
    ex = VecExplorer(env, env.robot, gamma=0.9, policy=object())
    res = ex.run_k_episodes(k, "test", action_fn=_goal_seeking, returnNav=True, total_envs=E_total)
    json.dump({"result": list(res), "records": ex.last_records}, open(out, "w"))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
