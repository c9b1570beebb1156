"""GPU: the sharded rollout on two ranks equals the single-process rollout.  Two fresh child processes (gloo,
both on GPU 0) each step half of the envs through the HIP path and meet in the one all_gather of episode records."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("E_total", [32, 33])          # 33: unequal shards (17 + 16 envs), sizes exchanged
def test_two_rank_vec_explorer_equals_single_process(tmp_path, E_total):
    from modelcrowdnav_amd.rollout import VecExplorer
    from tests import helpers as H
    from tests.test_rollout_gpu import _goal_seeking
    N, k = 5, 80
    env = H.make_vec_env(E_total, N)
    env.track_human_times = False
    env.export_human_actions = False
    ex = VecExplorer(env, env.robot, gamma=0.9, policy=object())
    want = list(ex.run_k_episodes(k, "test", action_fn=_goal_seeking, returnNav=True))
    want_rec = ex.last_records
    port = 30100 + (os.getpid() % 2000) + E_total % 7
    outs = [str(tmp_path / ("r%d.json" % r)) for r in range(2)]
    envv = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-m", "tests.dist_worker", str(r), "2", str(port), str(E_total), str(N),
                               str(k), outs[r]], cwd=ROOT, env=envv, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True) for r in range(2)]
    logs = []
    for pr in procs:
        try:
            o, _ = pr.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    assert all(pr.returncode == 0 for pr in procs), "\n".join(logs)
    for o in outs:
        got = json.load(open(o))
        assert got["result"] == want                               # every rank reports the whole job
        for key in ("returns", "infos", "times"):
            assert got["records"][key] == want_rec[key], key
    assert len(set(want_rec["infos"])) > 1


def _run_child(args, timeout=420, env_extra=None):
    envv = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    pr = subprocess.Popen(args, cwd=ROOT, env=envv, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        out, err = pr.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        pr.kill()
        raise
    assert pr.returncode == 0, out[-3000:] + "\n" + err[-3000:]
    return out


def test_rccl_world_size_one_gather_on_device_tensors(tmp_path):
    """RCCL itself: ONE fresh child process forms a world-size-1 `nccl` (= RCCL) group on the GPU and runs
    dist.gather_records on device tensors (the all_gather_into_tensor of the rollout path, incl. the size exchange
    of unequal shards) and a sharded VecExplorer rollout whose records come back through that collective."""
    out = str(tmp_path / "rccl.json")
    port = 31100 + (os.getpid() % 2000)
    _run_child([sys.executable, "-m", "tests.rccl_worker", str(port), out])
    got = json.load(open(out))
    assert got["backend"] == "nccl" and got["world"] == 1
    assert got["gather_equal_ok"] and got["gather_sizes_ok"] and got["records_on_cuda"]
    # the rollout through the RCCL collective equals the same rollout without a process group
    from modelcrowdnav_amd.rollout import VecExplorer
    from tests import helpers as H
    from tests.test_rollout_gpu import _goal_seeking
    env = H.make_vec_env(32, 5)
    env.track_human_times = False
    env.export_human_actions = False
    ex = VecExplorer(env, env.robot, gamma=0.9, policy=object())
    want = list(ex.run_k_episodes(80, "test", action_fn=_goal_seeking, returnNav=True))
    assert got["result"] == want
    for key in ("returns", "infos", "times"):
        assert got["records"][key] == ex.last_records[key], key


def test_bench_collective_branches_on_rccl():
    """bench.py's N > 1 branches (graph captured before init_process_group, communicator warm-up, the one
    all_gather_into_tensor of episode records, MAX / SUM all-reduces, barrier) executed on RCCL with one rank."""
    port = 33100 + (os.getpid() % 2000)
    out = _run_child([sys.executable, "bench.py", "--gpus", "1", "--steps", "20", "--warmup", "2", "--force-collective",
                      "--no-sweep", "--no-extra", "--no-cpu-baseline", "--min-timed-ms", "5"],
                     env_extra={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                                "MASTER_PORT": str(port)})
    line = json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["collective"] == "nccl"
    assert line["gathered_episode_records"] is not None and line["gathered_episode_records"] >= 0
    assert line["value"] > 1e6


def test_bench_two_ranks_through_the_drivers_launcher():
    """The driver's own command line for N = 2 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node 2
    --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...` -- rehearsed on this one-GPU box: both ranks share
    GPU 0 and the collectives run on gloo (MCN_BENCH_BACKEND), everything else is the path the 8-GPU node takes:
    rank / local-rank from the launcher's environment, one JSON line from rank 0, whole-job value over both ranks,
    episode records of both shards gathered."""
    port = 35100 + (os.getpid() % 2000)
    out = _run_child([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                      "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "20",
                      "--warmup", "5", "--no-sweep", "--no-extra", "--no-cpu-baseline", "--min-timed-ms", "5"],
                     timeout=600, env_extra={"MCN_BENCH_BACKEND": "gloo"})
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]                      # rank 0 only
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 20 and line["warmup"] == 5 and line["scaling"] == "weak"
    assert line["collective"] == "gloo" and line["config"]["envs_per_gpu"] == 4096
    assert "x2" in line["config"]["parallelism"]
    assert line["gathered_episode_records"] is not None and line["gathered_episode_records"] > 0
    assert line["value"] > 1e8 and line["cpu_baseline"] is None


def _dp_children(ws, backend, mode, tmp_path, tag):
    import torch
    port = 32100 + (os.getpid() % 1500) + (7 if mode == "g10" else 0) + (3 if backend == "nccl" else 0)
    outs = [str(tmp_path / ("%s_%d.pt" % (tag, r))) for r in range(ws)]
    envv = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, "-m", "tests.dp_worker", str(r), str(ws), str(port), backend, mode, outs[r]],
                              cwd=ROOT, env=envv, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(ws)]
    logs = []
    for pr in procs:
        try:
            o, _ = pr.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    assert all(pr.returncode == 0 for pr in procs), "\n".join(l[-3000:] for l in logs)
    return [torch.load(o, weights_only=True) for o in outs]


@pytest.mark.parametrize("ws,backend", [(2, "gloo"), (1, "nccl")])
def test_data_parallel_trainer_on_device_matches_reference_trainer(ws, backend, tmp_path, golden_dir):
    """f2 on the device with ranks (crowd_nav/utils/trainer.py:64-82 + one flat-bucket gradient all-reduce per step):
    the model, the memory shard and the gradient bucket live on cuda:0 in every rank.  The ranks split the memory of
    the REFERENCE Trainer's own recorded run (g10_trainer.npz, `batch` mode) row by row, start from different weights
    (sync_weights broadcasts rank 0's) and must end at the reference's weights: 2e-6, as the single-process device run.
    world 2: two child processes sharing GPU 0, the collective on host copies (gloo); world 1: the same code path through
    an `nccl` (= RCCL) group, i.e. the all-reduce runs on the device bucket itself."""
    import numpy as np
    res = _dp_children(ws, backend, "g10", tmp_path, "g10")
    g = np.load(os.path.join(golden_dir, "g10_trainer.npz"))
    assert all(r["backend"] == backend and r["world"] == ws for r in res)
    assert sum(r["rows"] for r in res) == g["batch_states"].shape[0]
    for r in res:
        for k, v in r["weights"].items():
            np.testing.assert_allclose(v.numpy(), g["batch_w1__" + k.replace(".", "__")], rtol=0, atol=2e-6, err_msg=k)
    for k in res[0]["weights"]:
        assert all(bool((r["weights"][k] == res[0]["weights"][k]).all()) for r in res), k        # ranks in lock step
    # each rank's loss is the MSE over its own rows; their row-weighted mean is the reference's batch loss
    n = [r["rows"] for r in res]
    for i in range(3):
        got = sum(r["losses"][i] * m for r, m in zip(res, n)) / sum(n)
        assert abs(got - float(g["batch_losses"][i])) < 2e-6, (i, got)


def test_data_parallel_trainer_on_device_unequal_shards(tmp_path):
    """Two ranks on GPU 0 with 32 vs 45 rows (2 vs 3 mini-batches of 16): the epoch's step count is agreed by one
    all-reduce(MAX), the short rank wraps around its permutation, and the weights stay identical on both ranks."""
    import torch
    res = _dp_children(2, "gloo", "unequal", tmp_path, "uneq")
    assert [r["rows"] for r in res] == [32, 45]
    from tests.test_training_cpu import _model
    init = _model(seed=0).state_dict()
    for k in res[0]["weights"]:
        assert torch.equal(res[0]["weights"][k], res[1]["weights"][k]), k
    assert not torch.equal(res[0]["weights"]["mlp1.0.weight"], init["mlp1.0.weight"])
