"""GPU: the batched model-based pipeline of INTEGRATION.md section 2 end to end on the synthetic recording:
ndjson ingest -> world-model training (Trainer_Sim) -> mixed-reality data generation with the robot replacing a
recorded pedestrian (VecDataGen) -> value-network training (Trainer).  Pins that the pieces compose; each piece has
its own parity test."""
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests import helpers as H  # noqa: E402


def test_recorded_data_to_value_network(golden_dir, tmp_path):
    import torch
    from modelcrowdnav_amd import configs
    from modelcrowdnav_amd.envs import VecModelCrowdSim
    from modelcrowdnav_amd.policy.sarl import SARL
    from modelcrowdnav_amd.policy.world_model import MlpWorld, VecMlpWorld, vec_world
    from modelcrowdnav_amd.utils import realdata
    from modelcrowdnav_amd.utils.datagen import VecDataGen
    from modelcrowdnav_amd.utils.memory import ReplayMemory
    from modelcrowdnav_amd.utils.trainer import Trainer
    from modelcrowdnav_amd.utils.trainer_sim import Trainer_Sim
    dev = torch.device("cuda", 0)
    torch.manual_seed(0); random.seed(0)
    data = realdata.get_real_data(os.path.join(golden_dir, "g9_scenes.ndjson"), phase="test", stride=1, windows_size=12,
                                  padding_last="moving", padding_first="stay", cache_dir=str(tmp_path))
    obs, lengths, ids = data.episode_tensor()
    n_h = obs.shape[2]
    # keep whole episodes of the majority pedestrian count
    keep_ids = set(ids)
    rows, cur = [], 0
    for sc in data.scenes:
        T = sc["obs"].shape[0]
        if sc["id"] in keep_ids:
            rows += data.raw_memory()[cur:cur + T]
        cur += T
    assert len(rows) == int(lengths.sum()) and os.listdir(tmp_path)

    pairs = [p for p in data.world_pairs() if p[0].shape[0] == n_h]
    world = MlpWorld(n_h - 1, drop_rate=0.0).to(dev)        # the robot replaces one pedestrian: n_h - 1 remain
    sub = [(p[0][:n_h - 1], p[1][:n_h - 1]) for p in pairs]
    wt = Trainer_Sim(world, sub, dev, 32, str(tmp_path / "world.pth"))
    wt.set_learning_rate(2e-3)
    l0 = wt.optimize_epoch(1, reset=True)
    l1 = wt.optimize_epoch(30)
    assert l1 <= l0 and world.mse == l1

    env = H.make_vec_env(16, n_h - 1, cls=VecModelCrowdSim)
    pol = SARL(); pol.configure(configs.policy_config()); pol.kinematics = "holonomic"
    pol.set_device(dev); pol.set_phase("train"); pol.set_epsilon(0.1); pol.time_step = 0.25
    env.robot.set_policy(pol); pol.set_env(env)
    env.sim_world = vec_world(world.eval(), env)          # the HIP kernel (mcn_mlp_world_step)
    assert isinstance(env.sim_world, VecMlpWorld)
    memory = ReplayMemory(50000, device=dev)
    gen = VecDataGen(memory, env.robot, env, pol)
    gen.raw_memory = rows
    out = gen.gen_data_from_explore_in_mix(40, phase="train", min_end=4, imitation_learning=True, add_sim=True,
                                           replace_robot=True, returnRate=False)
    assert out[1] + out[2] + out[3] == 40
    assert env._alloc_N == n_h - 1
    if len(memory):                                       # an untrained robot mostly times out; collisions feed the memory
        tr = Trainer(pol.get_model(), memory, dev, 50)
        tr.set_learning_rate(0.01)
        assert np.isfinite(tr.optimize_batch(5))


@pytest.mark.parametrize("N", [5, 3, 10, 8])
def test_mlp_world_kernel_matches_torch_module(N):
    """mcn_mlp_world_step (world_mlp.hip) against MlpWorld.forward in eval mode (crowd_nav/policy/world_model.py:22-42)
    on the input row model_crowd_sim.py:401-405 builds: float32 network, 1e-5 on the tanh outputs; ragged E."""
    import torch
    from modelcrowdnav_amd.envs import VecModelCrowdSim
    from modelcrowdnav_amd.policy.world_model import MlpWorld, VecMlpWorld, VecTorchWorld
    from tests import helpers as H
    E = 333
    dev = torch.device("cuda", 0)
    env = H.make_vec_env(E, N, cls=VecModelCrowdSim)
    rng = np.random.RandomState(N)
    H.upload(env, H.random_state(rng, E, N))
    torch.manual_seed(N)
    world = MlpWorld(N).to(dev).eval()
    with torch.no_grad():                       # spread the outputs over tanh's range
        for prm in world.parameters():
            prm.mul_(1.7)
    got = VecMlpWorld(world, env)(env.hpos).clone()
    want = VecTorchWorld(world, env)(env.hpos)
    assert got.shape == (E, N, 2) and got.dtype == torch.float64
    err = float((got - want).abs().max())
    assert err <= 1e-5, err
    assert float(want.abs().max()) > 0.3 and float(want.std()) > 0.05
    # a VecModelCrowdSim steps its humans with it
    env.sim_world = VecMlpWorld(world, env)
    before = env.hpos.clone()
    env.step(torch.zeros(E, 2, dtype=torch.float64, device=dev))
    moved = env.hpos - before
    live = ~env.done.bool()
    assert torch.allclose(moved[live], (got * env.time_step)[live], atol=1e-12)


@pytest.mark.parametrize("kind", ["mlp", "attention"])
def test_world_kernels_follow_the_module_through_training_steps(kind):
    """train_model_based.py alternates Trainer_Sim.optimize_epoch with imagination on the SAME module: after an optimizer
    step (and after load_state_dict) the HIP adapters must imagine with the new weights without a manual refresh(); and a
    MlpWorld left in train() mode (Dropout 0.5 active, as before the first optimize_epoch) must not silently get the
    kernel's eval-mode forward."""
    import torch
    from modelcrowdnav_amd.envs import VecModelCrowdSim
    from modelcrowdnav_amd.policy.world_model import AttentionWorld, MlpWorld, VecTorchWorld, vec_world
    from tests import helpers as H
    E, N = 200, 5
    dev = torch.device("cuda", 0)
    env = H.make_vec_env(E, N, cls=VecModelCrowdSim)
    H.upload(env, H.random_state(np.random.RandomState(2), E, N))
    torch.manual_seed(3)
    world = (MlpWorld(N) if kind == "mlp" else AttentionWorld()).to(dev).eval()
    fast, slow = vec_world(world, env), VecTorchWorld(world, env)
    assert type(fast).__name__ in ("VecMlpWorld", "VecAttnWorld")
    v0 = fast(env.hpos).clone()
    assert float((v0 - slow(env.hpos)).abs().max()) <= 1e-5
    # one Adam step on a made-up target (what Trainer_Sim.optimize_epoch does to the module, trainer_sim.py:60-75)
    opt = torch.optim.Adam(world.parameters(), lr=0.05)
    world.train()
    x = torch.cat([env.hpos, env.hvel], 2).reshape(E, -1).float()
    loss = (world(x) - 0.3).pow(2).mean()
    loss.backward()
    opt.step()
    world.eval()
    v1 = fast(env.hpos).clone()                     # no refresh()
    want1 = slow(env.hpos)
    # (AttentionWorld has no output non-linearity: after a large step its outputs are in the hundreds -- relative bar)
    assert float((v1 - want1).abs().max()) <= 1e-5 * max(1.0, float(want1.abs().max()))
    assert float((v1 - v0).abs().max()) > 1e-3, "the step must have changed the predictions"
    # load_state_dict replaces the values in place
    torch.manual_seed(9)
    other = (MlpWorld(N) if kind == "mlp" else AttentionWorld()).to(dev)
    world.load_state_dict(other.state_dict())
    assert float((fast(env.hpos) - slow(env.hpos)).abs().max()) <= 1e-5
    if kind == "mlp":
        world.train()                               # Dropout(0.5) draws masks now: not the kernel's function
        torch.manual_seed(4)
        a = fast(env.hpos).clone()
        torch.manual_seed(4)
        b = slow(env.hpos)
        assert torch.equal(a, b)                    # the adapter went through the module, same random stream
        world.eval()
        assert float((fast(env.hpos) - slow(env.hpos)).abs().max()) <= 1e-5


@pytest.mark.parametrize("N", [5, 1, 10, 7])
def test_attention_world_kernel_matches_torch_module(N):
    """mcn_attn_world_step (world_attn.hip) against AttentionWorld.forward (crowd_nav/policy/world_model.py:54-106):
    float32 network in another summation order (and with mlp2.2 / the pooled half of mlp3.0 applied once per scene),
    1e-5 on the predicted velocities; ragged E; per-scene pedestrian counts against the module run on the shorter
    scene."""
    import torch
    from modelcrowdnav_amd.envs import VecModelCrowdSim
    from modelcrowdnav_amd.policy.world_model import AttentionWorld, VecAttnWorld, VecTorchWorld
    from tests import helpers as H
    E = 203
    dev = torch.device("cuda", 0)
    env = H.make_vec_env(E, N, cls=VecModelCrowdSim)
    rng = np.random.RandomState(10 + N)
    H.upload(env, H.random_state(rng, E, N))
    torch.manual_seed(N)
    world = AttentionWorld().to(dev).eval()
    fast = VecAttnWorld(world, env)
    got = fast(env.hpos).clone()
    want = VecTorchWorld(world, env)(env.hpos)
    assert got.shape == (E, N, 2) and got.dtype == torch.float64
    err = float((got - want).abs().max())
    assert err <= 1e-5, err
    assert float(want.abs().max()) > 1e-2
    if N >= 3:
        counts = torch.from_numpy(rng.randint(1, N + 1, E).astype(np.int32)).to(dev)
        part = fast(env.hpos, hcount=counts).clone()
        x = torch.cat([env.hpos, env.hvel], dim=2).float()
        for e in range(0, E, 17):
            n = int(counts[e])
            with torch.no_grad():
                ref = world(x[e:e + 1, :n].reshape(1, -1)).view(n, 2).double()
            assert float((part[e, :n] - ref).abs().max()) <= 1e-5, e


@pytest.mark.parametrize("case", ["mlp1", "mlp5", "mlp10", "attn2", "attn5", "attn10"])
def test_world_kernels_match_reference_outputs(case, golden_dir):
    """mcn_mlp_world_step / mcn_attn_world_step against g13_world.npz = the REAL reference's MlpWorld (eval) and
    AttentionWorld forward (crowd_nav/policy/world_model.py:22-106) on 192 seeded scenes with seeded weights: float32
    networks in another summation order, 1e-5 on the predicted velocities (the float tolerance of the path)."""
    import torch
    from modelcrowdnav_amd.envs import VecModelCrowdSim
    from modelcrowdnav_amd.policy.world_model import AttentionWorld, MlpWorld, VecAttnWorld, VecMlpWorld, vec_world
    from tests import helpers as H
    g = np.load(os.path.join(golden_dir, "g13_world.npz"))
    is_mlp = case.startswith("mlp")
    N = int(case[3:] if is_mlp else case[4:])
    pref = case + "_w__" if is_mlp else "attn_w__"
    dev = torch.device("cuda", 0)
    m = MlpWorld(N) if is_mlp else AttentionWorld()
    m.load_state_dict({k[len(pref):].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith(pref)})
    m.to(dev).eval()
    x = g[case + "_in"].reshape(192, N, 4)
    env = H.make_vec_env(192, N, cls=VecModelCrowdSim)
    H.upload(env, H.random_state(np.random.RandomState(0), 192, N))         # allocates the state; overwritten below
    env.hpos.copy_(torch.from_numpy(x[:, :, :2].astype(np.float64)))
    env.hvel.copy_(torch.from_numpy(x[:, :, 2:].astype(np.float64)))
    world = vec_world(m, env)
    assert isinstance(world, VecMlpWorld if is_mlp else VecAttnWorld)
    got = world(env.hpos).cpu().numpy().reshape(192, 2 * N)
    err = np.abs(got - g[case + "_out"]).max()
    assert err <= 1e-5, err


def test_e1_model_crowd_sim_steps_with_a_torch_world_module():
    """The reference's E = 1 surface (model_crowd_sim.py:398-417): `env.sim_world = MlpWorld(...)` -- a plain nn.Module,
    not an SGANWorld -- gets the scene as one float32 row and its output row moves the humans."""
    import torch
    from modelcrowdnav_amd import configs
    from modelcrowdnav_amd.envs import ModelCrowdSim
    from modelcrowdnav_amd.envs.policy.policy_factory import policy_factory
    from modelcrowdnav_amd.envs.utils.action import ActionXY
    from modelcrowdnav_amd.envs.utils.robot import Robot
    from modelcrowdnav_amd.policy.world_model import MlpWorld
    cfg = configs.env_config()
    env = ModelCrowdSim()
    env.configure(cfg)
    robot = Robot(cfg, "robot")
    pol = policy_factory["orca"]()
    pol.multiagent_training = True
    robot.set_policy(pol)
    env.set_robot(robot)
    ob = env.reset("test", 3)
    N = len(ob)
    torch.manual_seed(1)
    world = MlpWorld(N).eval()
    env.sim_world = world
    before = np.array([[h.px, h.py, h.vx, h.vy] for h in env.humans])
    with torch.no_grad():
        want = world(torch.Tensor([before.tolist()]).reshape(1, -1))[0].reshape(N, 2).double().numpy()
    ob2, reward, done, info = env.step(ActionXY(0.0, 0.0))
    after = np.array([[h.px, h.py, h.vx, h.vy] for h in env.humans])
    assert np.allclose(after[:, 2:4], want, atol=0) and np.allclose(after[:, 0:2], before[:, 0:2] + want * env.time_step, atol=1e-15)
