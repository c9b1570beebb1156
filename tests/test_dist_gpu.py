"""GPU: the sharded rollout on two ranks equals the single-process rollout.  Two fresh child processes (gloo,
both on GPU 0) each step half of the envs through the HIP path and meet in the one all_gather of episode records."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_vec_explorer_equals_single_process(tmp_path):
    from modelcrowdnav_amd.rollout import VecExplorer
    from tests import helpers as H
    from tests.test_rollout_gpu import _goal_seeking
    E_total, N, k = 32, 5, 80
    env = H.make_vec_env(E_total, N)
    env.track_human_times = False
    env.export_human_actions = False
    ex = VecExplorer(env, env.robot, gamma=0.9, policy=object())
    want = list(ex.run_k_episodes(k, "test", action_fn=_goal_seeking, returnNav=True))
    want_rec = ex.last_records
    port = 30100 + (os.getpid() % 2000)
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
