"""GPU, end to end: the training procedure of crowd_nav/train.py:142-230 on batched envs -- imitation learning from
the ORCA robot (explorer.run_k_episodes(update_memory, imitation_learning) + trainer.optimize_epoch), target-network
update, then epsilon-greedy RL episodes (update_memory) + trainer.optimize_batch -- through VecExplorer, the HIP env
step / ORCA / SARL look-ahead kernels, the device replay memory and the Trainer.  No parity target here (training is
stochastic); the test pins that the pieces compose: the value loss falls, the imitation-trained SARL robot reaches
goals where the untrained one only times out, and RL samples keep feeding the memory."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from tests import helpers as H  # noqa: E402


def test_imitation_then_reinforcement_learning_loop():
    import torch
    from modelcrowdnav_amd import configs
    from modelcrowdnav_amd.envs.policy.policy_factory import policy_factory
    from modelcrowdnav_amd.policy.sarl import SARL
    from modelcrowdnav_amd.rollout import VecExplorer
    from modelcrowdnav_amd.utils.memory import ReplayMemory
    from modelcrowdnav_amd.utils.trainer import Trainer
    dev = torch.device("cuda", 0)
    E, N = 256, 5
    env = H.make_vec_env(E, N)
    env.track_human_times = False; env.export_human_actions = False
    torch.manual_seed(0)
    sarl = SARL(); sarl.configure(configs.policy_config()); sarl.kinematics = "holonomic"
    sarl.multiagent_training = True
    sarl.set_device(dev); sarl.set_phase("train"); sarl.time_step = env.time_step
    model = sarl.get_model()
    memory = ReplayMemory(100000, device=dev)
    trainer = Trainer(model, memory, dev, 100)

    # the untrained network: evaluation episodes all time out (nothing pulls the robot to its goal)
    env.robot.set_policy(sarl)
    ex = VecExplorer(env, env.robot, gamma=0.9, policy=sarl, memory=memory)
    sarl.set_epsilon(0.0)
    _, sr0, _, _ = ex.run_k_episodes(E, "val")
    assert sr0 < 0.2

    # ---- imitation learning (train.py:142-176): ORCA robot with safety space, IL value targets, epochs of SGD
    orca = policy_factory["orca"]()
    orca.multiagent_training = True
    orca.safety_space = 0.15
    env.robot.set_policy(orca)
    il = VecExplorer(env, env.robot, gamma=0.9, policy=orca, memory=memory, target_policy=sarl)
    il.run_k_episodes(3 * E, "train", update_memory=True, imitation_learning=True)
    n_il = len(memory)
    assert n_il > 3 * E * 20                                   # tens of steps per successful episode
    trainer.set_learning_rate(0.01)                            # train.config imitation_learning.il_learning_rate
    first = trainer.optimize_epoch(1)
    last = trainer.optimize_epoch(12)
    assert last < 0.6 * first, (first, last)

    # ---- the imitation-trained robot now drives itself (greedy evaluation, batched look-ahead kernel)
    env.robot.set_policy(sarl)
    sarl.set_phase("val"); sarl.set_epsilon(0.0)
    _, sr1, cr1, _ = ex.run_k_episodes(E, "val")
    assert sr1 > sr0 + 0.3, (sr0, sr1, cr1)

    # ---- reinforcement learning (train.py:178-230): target network, epsilon-greedy samples, batches of SGD
    ex.update_target_model(model)
    trainer.set_learning_rate(0.001)
    sarl.set_phase("train"); sarl.set_epsilon(0.3)
    before = memory.position
    _, success, collision, timeout = ex.run_k_episodes(E, "train", update_memory=True, episode=0, returnRate=False)
    assert success + collision + timeout == E and success > 0
    assert memory.position != before                          # RL experience went into the ring
    loss = trainer.optimize_batch(50)
    assert np.isfinite(loss) and loss < 1.0
    ex.update_target_model(model)
    sarl.set_phase("val"); sarl.set_epsilon(0.0)
    _, sr2, _, _ = ex.run_k_episodes(E, "val")
    assert sr2 > sr0 + 0.3


@pytest.mark.parametrize("mode", ["batch", "epoch"])
def test_trainer_steps_match_reference_trainer_on_device(mode, golden_dir):
    """SURVEY 8f f2 on the device: utils/trainer.py with model, replay memory and optimiser on cuda:0 against the
    REAL reference's Trainer (crowd_nav/utils/trainer.py:64-82; g10_trainer.npz).  Tolerance 2e-6 on weights and
    losses (the CPU run holds 1e-6; rocBLAS sums the float32 dot products in another order)."""
    import torch
    from tests.test_training_cpu import check_trainer_against_reference
    worst = check_trainer_against_reference(mode, golden_dir, torch.device("cuda", 0), 2e-6)
    print("trainer on device, mode %s: max |w - w_ref| = %.3g" % (mode, worst))


@pytest.mark.parametrize("mode", ["rl", "il"])
def test_value_targets_match_reference_update_memory_on_device(mode, golden_dir):
    """rollout.value_targets on cuda:0 against the reference's Explorer.update_memory (explorer.py:153-186;
    g12_update_memory.npz): stored states exact, values to 2e-6."""
    import torch
    from tests.test_training_cpu import check_value_targets_against_reference
    worst = check_value_targets_against_reference(mode, golden_dir, torch.device("cuda", 0), 2e-6)
    print("value targets on device, mode %s: max |v - v_ref| = %.3g" % (mode, worst))
