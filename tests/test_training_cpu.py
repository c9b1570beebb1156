"""CPU: the next-tier callers -- replay memory ring semantics, trainer steps, and the
data-parallel gradient all-reduce on world_size-2 gloo."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(seed=0):
    from modelcrowdnav_amd import configs
    from modelcrowdnav_amd.policy.sarl import SARL
    torch.manual_seed(seed)
    p = SARL()
    p.configure(configs.policy_config())
    return p.model


def test_replay_memory_ring_matches_reference_semantics():
    from modelcrowdnav_amd.utils.memory import ReplayMemory
    m = ReplayMemory(5)
    items = [(torch.full((3, 13), float(i)), torch.tensor([float(i)])) for i in range(8)]
    for it in items[:3]:
        m.push(it)
    assert len(m) == 3 and not m.is_full() and m.position == 3
    for it in items[3:]:
        m.push(it)                      # wraps: slots 0..2 now hold items 5,6,7 (memory.py:13-19)
    assert len(m) == 5 and m.is_full() and m.position == 3
    assert [float(m[i][1]) for i in range(5)] == [5.0, 6.0, 7.0, 3.0, 4.0]
    m.push_batch(torch.stack([it[0] for it in items[:4]]), torch.tensor([10.0, 11.0, 12.0, 13.0]))
    assert [float(m[i][1]) for i in range(5)] == [12.0, 13.0, 7.0, 10.0, 11.0] and m.position == 2
    s, v = m.sample(4, torch.Generator().manual_seed(0))
    assert s.shape == (4, 3, 13) and v.shape == (4, 1)
    m.clear()
    assert len(m) == 0


def test_trainer_reduces_loss():
    from modelcrowdnav_amd.utils.memory import ReplayMemory
    from modelcrowdnav_amd.utils.trainer import Trainer
    model = _model()
    g = torch.Generator().manual_seed(1)
    states = torch.randn(200, 5, 13, generator=g)
    states[:, :, :6] = states[:, :1, :6]
    values = torch.tanh(states[:, :, 6].mean(1))
    mem = ReplayMemory(400)
    mem.push_batch(states, values)
    tr = Trainer(model, mem, torch.device("cpu"), 100)
    try:
        tr.optimize_batch(1)
        assert False, "learning rate must be set first"
    except ValueError:
        pass
    tr.set_learning_rate(0.01)
    first = tr.optimize_epoch(1)
    for _ in range(10):
        last = tr.optimize_epoch(1)
    assert last < first
    assert tr.optimize_batch(3) > 0


def _load_sd(model, g, prefix):
    sd = {k: torch.from_numpy(g[prefix + k.replace(".", "__")].copy()) for k in model.state_dict().keys()}
    model.load_state_dict(sd)


def check_trainer_against_reference(mode, golden_dir, device, atol):
    """utils/trainer.py against the REAL reference Trainer (crowd_nav/utils/trainer.py:19-82; fixture
    tests/golden/g10_trainer.npz from tests/golden_tools/gen_golden_nets.py:g10_trainer): same seeded ValueNetwork,
    same memory rows, lr 0.01; three optimize_batch(1) calls over a one-batch memory / two optimize_epoch(1) calls over a
    two-batch memory with the DataLoader's recorded permutations.  Weights to 1e-6, losses to 1e-6."""
    from modelcrowdnav_amd.utils.memory import ReplayMemory
    from modelcrowdnav_amd.utils.trainer import Trainer
    g = np.load(os.path.join(golden_dir, "g10_trainer.npz"))
    torch.set_num_threads(1)
    model = _model(seed=3)
    _load_sd(model, g, "w0__")
    states, values = torch.from_numpy(g[mode + "_states"]), torch.from_numpy(g[mode + "_values"])
    model.to(device)
    mem = ReplayMemory(states.shape[0], device=device) if device.type != "cpu" else ReplayMemory(states.shape[0])
    mem.push_batch(states.to(device), values.reshape(-1).to(device))
    tr = Trainer(model, mem, device, 100)
    tr.set_learning_rate(0.01)
    if mode == "batch":
        losses = [tr.optimize_batch(1) for _ in range(3)]
    else:
        losses = [tr.optimize_epoch(1, perms=g["epoch_perms"][i:i + 1]) for i in range(2)]
    np.testing.assert_allclose(losses, g[mode + "_losses"], rtol=0, atol=atol)
    worst = 0.0
    for k, v in model.state_dict().items():
        want = g[mode + "_w1__" + k.replace(".", "__")]
        np.testing.assert_allclose(v.cpu().numpy(), want, rtol=0, atol=atol, err_msg=k)
        worst = max(worst, float(np.abs(v.cpu().numpy() - want).max()))
    return worst


@pytest.mark.parametrize("mode", ["batch", "epoch"])
def test_trainer_steps_match_reference_trainer(mode, golden_dir):
    check_trainer_against_reference(mode, golden_dir, torch.device("cpu"), 1e-6)


def check_value_targets_against_reference(mode, golden_dir, device, atol):
    """rollout.value_targets (the batched Explorer.update_memory / DataGen.update_memory) against the REAL reference's
    Explorer.update_memory (explorer.py:153-186; fixture tests/golden/g12_update_memory.npz): three episodes of 7 / 12 /
    1 steps laid side by side as three envs, RL targets through a seeded target network, IL discounted tail sums."""
    from modelcrowdnav_amd import _hip
    from modelcrowdnav_amd.rollout import value_targets
    g = np.load(os.path.join(golden_dir, "g12_update_memory.npz"))
    torch.set_num_threads(1)
    model = _model(seed=5)
    _load_sd(model, g, "w__")
    lens = [len(g["%s_e%d_rewards" % (mode, e)]) for e in range(3)]
    T, E = max(lens) + 2, 3                       # two junk steps after the longest episode: must not be kept
    rng = np.random.RandomState(1)
    states = torch.from_numpy(rng.normal(0, 1, (T, E, 5, 13)).astype(np.float32))
    rewards = torch.from_numpy(rng.uniform(-1, 1, (T, E)))
    dones = torch.zeros(T, E, dtype=torch.bool)
    infos = torch.zeros(T, E, dtype=torch.uint8)
    for e, L in enumerate(lens):
        states[:L, e] = torch.from_numpy(g["%s_e%d_mem_states" % (mode, e)])
        rewards[:L, e] = torch.from_numpy(g["%s_e%d_rewards" % (mode, e)])
        dones[L - 1, e] = True
        infos[L - 1, e] = _hip.INFO_REACHGOAL if e != 1 else _hip.INFO_COLLISION
    gbar = pow(0.9, 0.25 * 1.0)
    model.to(device)
    s, v = value_targets(states.to(device), rewards.to(device), dones.to(device), infos.to(device), mode == "il", gbar,
                         target_model=model, device=device)
    want_v = np.concatenate([g["%s_e%d_values" % (mode, e)] for e in range(3)])
    want_s = np.concatenate([g["%s_e%d_mem_states" % (mode, e)] for e in range(3)])
    assert tuple(s.shape) == want_s.shape and np.array_equal(s.cpu().numpy(), want_s)   # (env, time) order = episode order
    np.testing.assert_allclose(v.cpu().numpy(), want_v, rtol=0, atol=atol)
    return float(np.abs(v.cpu().numpy() - want_v).max())


@pytest.mark.parametrize("mode", ["rl", "il"])
def test_value_targets_match_reference_update_memory(mode, golden_dir):
    check_value_targets_against_reference(mode, golden_dir, torch.device("cpu"), 1e-6)


def _dp_unequal_worker(rank, ws, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(ws), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from modelcrowdnav_amd import dist as mdist
    from modelcrowdnav_amd.utils.memory import ReplayMemory
    from modelcrowdnav_amd.utils.trainer import Trainer
    mdist.init_from_env("gloo")
    torch.set_num_threads(1)
    model = _model(seed=0)
    g = torch.Generator().manual_seed(11)
    states = torch.randn(77, 5, 13, generator=g)
    values = torch.randn(77, generator=g)
    rows = slice(0, 32) if rank == 0 else slice(32, 77)          # 32 vs 45 rows: 2 vs 3 mini-batches of 16
    mem = ReplayMemory(64)
    mem.push_batch(states[rows], values[rows])
    tr = Trainer(model, mem, torch.device("cpu"), 16)
    tr.sync_weights()
    tr.set_learning_rate(0.02)
    tr.optimize_epoch(3)                                         # would hang if the ranks disagreed on the step count
    tr.optimize_batch(2)
    torch.save({k: v.clone() for k, v in model.state_dict().items()}, os.path.join(out_dir, "u%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_trainer_unequal_shards(tmp_path):
    """Ranks with different amounts of experience (32 vs 45 rows) agree on the number of steps per epoch and stay
    in lock step (a differing count of all-reduces would deadlock: the spawn below would time out)."""
    port = 31300 + (os.getpid() % 1500)
    mp.spawn(_dp_unequal_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    w0 = torch.load(os.path.join(str(tmp_path), "u0.pt"), weights_only=True)
    w1 = torch.load(os.path.join(str(tmp_path), "u1.pt"), weights_only=True)
    for k in w0:
        assert torch.equal(w0[k], w1[k]), k
    assert not torch.equal(w0["mlp1.0.weight"], _model(seed=0).state_dict()["mlp1.0.weight"])


def _dp_worker(rank, ws, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(ws), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from modelcrowdnav_amd import dist as mdist
    from modelcrowdnav_amd.utils.memory import ReplayMemory
    from modelcrowdnav_amd.utils.trainer import Trainer
    mdist.init_from_env("gloo")
    torch.set_num_threads(1)
    model = _model(seed=rank)                   # deliberately different: sync_weights must fix it
    g = torch.Generator().manual_seed(7)
    states = torch.randn(64, 5, 13, generator=g)
    values = torch.randn(64, generator=g)
    mem = ReplayMemory(32)
    mem.push_batch(states[rank * 32:(rank + 1) * 32], values[rank * 32:(rank + 1) * 32])
    tr = Trainer(model, mem, torch.device("cpu"), 32)
    tr.sync_weights()
    tr.set_learning_rate(0.05)
    tr.optimize_epoch(2)                        # batch == whole shard, so the order inside it is irrelevant
    torch.save({k: v.clone() for k, v in model.state_dict().items()}, os.path.join(out_dir, "w%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_trainer_world2(tmp_path):
    from modelcrowdnav_amd.utils.memory import ReplayMemory
    from modelcrowdnav_amd.utils.trainer import Trainer
    port = 29800 + (os.getpid() % 1500)
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    w0 = torch.load(os.path.join(str(tmp_path), "w0.pt"), weights_only=True)
    w1 = torch.load(os.path.join(str(tmp_path), "w1.pt"), weights_only=True)
    for k in w0:
        assert torch.equal(w0[k], w1[k]), k                      # ranks stay in lock step
    # single process on the union with the same initial weights (rank 0's seed) must match:
    # mean over 64 samples == average of the two 32-sample means
    torch.set_num_threads(1)
    model = _model(seed=0)
    g = torch.Generator().manual_seed(7)
    states = torch.randn(64, 5, 13, generator=g)
    values = torch.randn(64, generator=g)
    mem = ReplayMemory(64)
    mem.push_batch(states, values)
    tr = Trainer(model, mem, torch.device("cpu"), 64)
    tr.set_learning_rate(0.05)
    tr.optimize_epoch(2)
    for k, v in model.state_dict().items():
        assert torch.allclose(v, w0[k], rtol=0, atol=2e-6), k


def test_world_model_modules_match_reference_parameter_names():
    """MlpWorld / AttentionWorld keep the reference's state_dict keys (world_model.py:22-106) and shapes."""
    from modelcrowdnav_amd.policy.world_model import MlpWorld, AttentionWorld
    m = MlpWorld(5)
    assert list(m.state_dict().keys()) == ["mlp.0.weight", "mlp.0.bias", "mlp.3.weight", "mlp.3.bias",
                                           "mlp.6.weight", "mlp.6.bias", "mlp.8.weight", "mlp.8.bias"]
    m.eval()
    assert m(torch.zeros(3, 20)).shape == (3, 10)
    a = AttentionWorld()
    keys = list(a.state_dict().keys())
    assert keys[0] == "mlp1.0.weight" and "attention.4.bias" in keys and "mlp3.6.weight" in keys
    assert a.state_dict()["mlp3.0.weight"].shape == (150, 54) and a.state_dict()["mlp3.6.weight"].shape == (2, 100)
    out = a(torch.randn(4, 20))
    assert out.shape == (4, 10) and a.attention_weights.shape == (5,)


@pytest.mark.parametrize("case", ["mlp1", "mlp5", "mlp10", "attn2", "attn5", "attn10"])
def test_world_model_modules_match_reference_outputs(case, golden_dir):
    """g13_world.npz = the reference's own MlpWorld (eval) / AttentionWorld forward (world_model.py:22-106) on seeded
    scenes: this repo's modules, loaded with the same weights, give the same rows (same torch ops on the same CPU;
    1e-6 leaves room for a differently blocked GEMM) and the same scene-0 attention weights."""
    from modelcrowdnav_amd.policy.world_model import MlpWorld, AttentionWorld
    g = np.load(os.path.join(golden_dir, "g13_world.npz"))
    N = int(case[3:] if case.startswith("mlp") else case[4:])
    pref = case + "_w__" if case.startswith("mlp") else "attn_w__"
    m = MlpWorld(N) if case.startswith("mlp") else AttentionWorld()
    m.load_state_dict({k[len(pref):].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith(pref)})
    m.eval()
    with torch.no_grad():
        y = m(torch.from_numpy(g[case + "_in"])).numpy()
    want = g[case + "_out"]
    assert y.shape == want.shape == (192, 2 * N)
    assert np.abs(y - want).max() <= 1e-6
    assert np.abs(want).max() > (0.05 if case.startswith("mlp") else 0.01)
    if case.startswith("attn"):
        np.testing.assert_allclose(np.asarray(m.attention_weights), g[case + "_weights0"], rtol=0, atol=1e-6)


def test_world_model_trainer_matches_reference_trainer(golden_dir, tmp_path):
    """g20_trainer_sim.npz = the reference's own Trainer_Sim (trainer_sim.py:26-110) on a seeded AttentionWorld (no
    dropout): `random.shuffle(memory)` (Python's generator, seeded; in place, so a second call splits what the first
    left) decides the 80 / 20 split, the DataLoaders' row orders are the recorded ones.  5 + 3 epochs (the second call
    keeps the best score), and a 30-row run whose validation rows carry the negated law, so that it stops after
    patience + 1 epochs with the first epoch's weights: best validation
    losses, `model.mse`, early-stopping state and the restored best weights."""
    import random
    from modelcrowdnav_amd.policy.world_model import AttentionWorld
    from modelcrowdnav_amd.utils.trainer_sim import Trainer_Sim
    g = np.load(os.path.join(golden_dir, "g20_trainer_sim.npz"))
    load = lambda pref: {k[len(pref):].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith(pref)}

    class Memory(object):                      # what the trainer needs of crowd_nav.utils.memory.ReplayMemory
        def __init__(self):
            self.memory = []

        def __len__(self):
            return len(self.memory)
    RT = {"short": 2e-5, "stop": 2e-5}
    for tag, calls, lr in (("short", (5, 3), 1e-3), ("stop", (50,), 1e-3)):
        model = AttentionWorld()
        model.load_state_dict(load("w0__"))
        mem = Memory()
        nxt = g["next"] if tag == "short" else g["next_stop"]      # "stop": the validation rows carry the negated law
        for i in range(int(g[tag + "_rows"])):
            mem.memory.append((torch.from_numpy(g["cur"][i]), torch.from_numpy(nxt[i])))
        tr = Trainer_Sim(model, mem, torch.device("cpu"), 1000, str(tmp_path / (tag + ".pt")))
        tr.set_learning_rate(lr)
        random.seed(2000)
        bests = [tr.optimize_epoch(e, perms=(g["%s_call%d_train_perms" % (tag, c)], g["%s_call%d_val_perms" % (tag, c)]))
                 for c, e in enumerate(calls)]
        # same ops on the same rows in the same order on the same kind of machine: 1e-6 relative leaves room for a
        # differently blocked GEMM
        np.testing.assert_allclose(bests, g[tag + "_best"], rtol=RT[tag], atol=0)
        assert abs(model.mse - float(g[tag + "_mse"])) <= RT[tag] * float(g[tag + "_mse"])
        assert tr.early_stopping.counter == int(g[tag + "_counter"])
        assert bool(tr.early_stopping.early_stop) == bool(g[tag + "_stopped"])
        want = load(tag + "_w1__")
        assert len(want) >= 6
        sd = model.state_dict()
        worst = max(float((sd[k] - v).abs().max()) for k, v in want.items())
        # Adam turns a gradient that is only rounding noise (dead ReLU units, the near-uniform attention branch) into
        # steps of size lr: single weights are determined to lr x steps, the function they compute much better
        assert worst <= lr * sum(calls), (tag, worst)
        if tag == "short":
            ref = AttentionWorld()
            ref.load_state_dict(want)
            x = torch.from_numpy(g["cur"][:64]).reshape(64, -1)
            with torch.no_grad():
                d = float((model(x) - ref(x)).abs().max())
            assert d <= 1e-4, d
    assert bool(g["stop_stopped"]) and int(g["stop_counter"]) == 7 and len(g["short_best"]) == 2


def test_world_model_trainer_early_stopping_and_best_weights(tmp_path):
    """Trainer_Sim (trainer_sim.py:26-110): fits MlpWorld to a synthetic 'next velocity = damped current velocity'
    law, returns the best validation loss, restores the best weights and records model.mse."""
    import random
    from modelcrowdnav_amd.policy.world_model import MlpWorld
    from modelcrowdnav_amd.utils.trainer_sim import Trainer_Sim, EarlyStopping
    torch.manual_seed(0); random.seed(0)
    N = 3
    g = torch.Generator().manual_seed(2)
    pairs = []
    for _ in range(600):
        cur = torch.rand(N, 4, generator=g) - 0.5
        pairs.append((cur, 0.8 * cur[:, 2:4]))
    pairs.append((torch.zeros(N + 1, 4), torch.zeros(N + 1, 2)))        # odd-sized pair: dropped like collate_fn does
    model = MlpWorld(N, drop_rate=0.0)
    path = str(tmp_path / "world.pth")
    tr = Trainer_Sim(model, pairs, torch.device("cpu"), 64, path)
    with pytest.raises(ValueError):
        tr.optimize_epoch(1)
    tr.set_learning_rate(3e-3)
    first = tr.optimize_epoch(1, reset=True)
    best = tr.optimize_epoch(40)
    assert best < 0.5 * first and abs(model.mse - best) < 1e-12
    saved = torch.load(path, weights_only=True)
    for k, v in model.state_dict().items():
        assert torch.equal(v, saved[k])                                  # the best weights are the ones in the model
    es = EarlyStopping(patience=2)
    lin = torch.nn.Linear(1, 1)
    for loss in (1.0, 0.9, 0.95, 0.93):
        es(loss, lin)
    assert es.early_stop and es.best_score == -0.9 and es.counter == 2
