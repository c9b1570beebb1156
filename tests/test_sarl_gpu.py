"""GPU parity of the fused SARL look-ahead (mcn_sarl_lookahead through the C ABI).

Tolerance: the network is float32 in the reference (torch CPU); the kernel evaluates it with
float32 MFMA (exact fmaf chains) in a different summation order, so values agree to ~1e-6.
The test bar is 1e-5 (BASELINE.json north_star) on values; the chosen action must be identical
whenever the top-2 gap exceeds that tolerance."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import pyref  # noqa: E402
from tests import helpers as H  # noqa: E402

TOL = 1e-5


def _policy(weights=None, seed=None):
    import torch
    from modelcrowdnav_amd import configs
    from modelcrowdnav_amd.policy.sarl import SARL
    if seed is not None:
        torch.manual_seed(seed)
    p = SARL()
    p.configure(configs.policy_config())
    p.kinematics = "holonomic"
    if weights is not None:
        p.model.load_state_dict(weights)
    p.set_device(torch.device("cuda", 0))
    p.set_phase("test")
    p.time_step = 0.25
    return p


def _weights(g, prefix):
    import torch
    return {k[len(prefix):].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith(prefix)}


def test_predict_matches_reference_fixture(golden_dir):
    """SARL.predict(JointState) against action_values / chosen action recorded from the real reference."""
    from modelcrowdnav_amd.envs.utils.state import FullState, ObservableState, JointState
    g = np.load(os.path.join(golden_dir, "g5_sarl.npz"))
    for seed in (0, 1):
        pol = _policy(_weights(g, "w%d__" % seed))
        for N in (5, 10):
            key = "pred%d_N%d_" % (seed, N)
            for s in range(g[key + "self"].shape[0]):
                me = FullState(*g[key + "self"][s].tolist())
                hs = [ObservableState(*row) for row in g[key + "humans"][s].tolist()]
                act = pol.predict(JointState(me, hs))
                want_vals, want_act = g[key + "values"][s], g[key + "action"][s]
                if np.isnan(want_vals[0]):
                    assert tuple(act) == (0, 0)
                    continue
                got = np.array(pol.action_values)
                np.testing.assert_allclose(got, want_vals, rtol=0, atol=TOL)
                top2 = np.sort(want_vals)[-2:]
                if top2[1] - top2[0] > 2 * TOL:
                    assert tuple(act) == tuple(want_act), (seed, N, s)


@pytest.mark.parametrize("N", [5, 10, 1, 3])
def test_predict_batch_matches_oracle(N):
    """predict_batch over a VecCrowdSim against the torch-fp32 restatement, env by env."""
    import torch
    rng = np.random.RandomState(N)
    E = 37          # ragged vs the 16-pair wave tile
    pol = _policy(seed=3)
    env = H.make_vec_env(E, N)
    st = H.random_state(rng, E, N, randomize=True)
    st.rgx[0], st.rgy[0] = st.rpx[0] + 0.05, st.rpy[0] - 0.05        # env 0: robot already at its goal
    H.upload(env, st)
    actions, best, values = pol.predict_batch(env, want_values=True)
    torch.cuda.synchronize()
    values, best, actions = values.cpu().numpy(), best.cpu().numpy(), actions.cpu().numpy()
    w = {k: v.detach().cpu() for k, v in pol.model.state_dict().items()}
    table = pol._action_table
    assert best[0] == -1 and tuple(actions[0]) == (0.0, 0.0)
    for e in range(1, E, 3):
        self_row = [st.rpx[e], st.rpy[e], st.rvx[e], st.rvy[e], st.rr[e], st.rgx[e], st.rgy[e], 1.0, 0.0]
        hum = np.stack([st.hpx[e], st.hpy[e], st.hvx[e], st.hvy[e], st.hr[e]], 1)
        ref, idx = pyref.sarl_predict(w, self_row, hum, table)
        np.testing.assert_allclose(values[e], ref, rtol=0, atol=TOL)
        top2 = np.sort(ref)[-2:]
        reached = float(np.linalg.norm((st.rpy[e] - st.rgy[e], st.rpx[e] - st.rgx[e]))) < st.rr[e]
        if reached:                       # policy.py:43-49 short cut: zero action, nothing evaluated
            assert best[e] == -1 and tuple(actions[e]) == (0.0, 0.0)
            continue
        assert best[e] == int(np.argmax(values[e]))       # first-max-wins on the device's own values
        if top2[1] - top2[0] > 2 * TOL:
            assert best[e] == idx and tuple(actions[e]) == tuple(table[idx])


@pytest.mark.parametrize("N", [5, 10])
def test_predict_batch_at_benchmark_size_matches_oracle(N):
    """The grids bench.py times (BASELINE configs 3 / 4: 4096 envs x 81 actions = 20 736 16-pair tiles, ten rounds
    of the resident workgroups): 64 sampled envs -- first, last, the tile-straddling ones and a random spread --
    against the torch-fp32 restatement (multi_human_rl.py:35-63, sarl.py:28-65)."""
    import torch
    rng = np.random.RandomState(40 + N)
    E = 4096
    pol = _policy(seed=3)
    env = H.make_vec_env(E, N)
    st = H.random_state(rng, E, N, randomize=True)
    H.upload(env, st)
    actions, best, values = pol.predict_batch(env, want_values=True)
    torch.cuda.synchronize()
    values, best, actions = values.cpu().numpy(), best.cpu().numpy(), actions.cpu().numpy()
    assert values.shape == (E, 81) and np.isfinite(values).all()
    w = {k: v.detach().cpu() for k, v in pol.model.state_dict().items()}
    table = pol._action_table
    sample = sorted(set([0, 1, 15, 16, 2047, 2048, E - 2, E - 1] + rng.choice(E, 56, replace=False).tolist()))
    worst = 0.0
    for e in sample:
        row = [st.rpx[e], st.rpy[e], st.rvx[e], st.rvy[e], st.rr[e], st.rgx[e], st.rgy[e], 1.0, 0.0]
        hum = np.stack([st.hpx[e], st.hpy[e], st.hvx[e], st.hvy[e], st.hr[e]], 1)
        reached = float(np.linalg.norm((st.rpy[e] - st.rgy[e], st.rpx[e] - st.rgx[e]))) < st.rr[e]
        if reached:
            assert best[e] == -1 and tuple(actions[e]) == (0.0, 0.0)
            continue
        ref, idx = pyref.sarl_predict(w, row, hum, table)
        np.testing.assert_allclose(values[e], ref, rtol=0, atol=TOL, err_msg="env %d" % e)
        worst = max(worst, float(np.abs(values[e] - ref).max()))
        assert best[e] == int(np.argmax(values[e]))
        top2 = np.sort(ref)[-2:]
        if top2[1] - top2[0] > 2 * TOL:
            assert best[e] == idx and tuple(actions[e]) == tuple(table[idx])
    # every env's action is the table row of its own argmax (the whole batch, not only the sample)
    moving = best >= 0
    assert np.array_equal(best[moving], np.argmax(values[moving], 1))
    assert np.array_equal(actions[moving], table[best[moving]])
    print("SARL 4096 x %d: max |value - reference| over %d sampled envs = %.3g" % (N, len(sample), worst))


def test_epsilon_greedy_in_train_phase():
    """multi_human_rl.py:27-29 per env: phase 'train' with epsilon = 1 replaces every choice by a uniformly drawn
    table action (except robots already on their goal, which return before the draw), epsilon = 0 is the greedy
    choice, and test phase ignores epsilon.  VecExplorer's training rollouts go through the same call."""
    import torch
    rng = np.random.RandomState(12)
    E, N = 2048, 5
    pol = _policy(seed=3)
    env = H.make_vec_env(E, N)
    st = H.random_state(rng, E, N)
    st.rgx[0], st.rgy[0] = st.rpx[0] + 0.05, st.rpy[0] - 0.05        # env 0: robot already at its goal
    H.upload(env, st)
    greedy_a, greedy_b = (x.clone() for x in pol.predict_batch(env))     # the outputs are the policy's own buffers
    pol.set_epsilon(1.0)
    a_test, b_test = pol.predict_batch(env)                          # phase 'test': epsilon is not looked at
    assert torch.equal(b_test, greedy_b) and torch.equal(a_test, greedy_a)
    pol.set_phase("train")
    torch.manual_seed(5)
    a1, b1 = (x.clone() for x in pol.predict_batch(env))
    assert int(b1[0]) == -1 and tuple(a1[0].tolist()) == (0.0, 0.0)
    moving = greedy_b >= 0
    assert bool((b1[moving] == -2).all())
    table = pol._bufs["table"]
    # every explored action is a table row, and the draw is spread over the table
    match = (a1[moving].unsqueeze(1) == table.unsqueeze(0)).all(2)
    assert bool(match.any(1).all())
    counts = match.float().sum(0)
    assert int((counts > 0).sum()) >= 75 and float(counts.max()) < 0.05 * float(moving.sum())
    assert float((a1[moving] != greedy_a[moving]).any(1).float().mean()) > 0.9
    pol.set_epsilon(0.25)
    a2, b2 = (x.clone() for x in pol.predict_batch(env))
    frac = float((b2[moving] == -2).float().mean())
    assert 0.2 < frac < 0.3
    keep = moving & (b2 != -2)
    assert torch.equal(a2[keep], greedy_a[keep]) and torch.equal(b2[keep], greedy_b[keep])
    pol.set_epsilon(0.0)
    a3, b3 = pol.predict_batch(env)
    assert torch.equal(b3, greedy_b) and torch.equal(a3, greedy_a)
    # reproducible under torch.manual_seed, different from call to call
    pol.set_epsilon(0.5)
    torch.manual_seed(9)
    x1 = pol.predict_batch(env)[0].clone()
    x2 = pol.predict_batch(env)[0].clone()
    torch.manual_seed(9)
    y1 = pol.predict_batch(env)[0].clone()
    assert torch.equal(x1, y1) and not torch.equal(x1, x2)


def test_rewards_are_float64_exact():
    """With a zeroed network V == 0, so values are exactly MultiHumanRL.compute_reward (float64)."""
    import torch
    from oracle import cport
    rng = np.random.RandomState(9)
    E, N = 200, 5
    pol = _policy(seed=0)
    with torch.no_grad():
        for p_ in pol.model.parameters():
            p_.zero_()
    env = H.make_vec_env(E, N)
    st = H.random_state(rng, E, N, crowded_frac=0.6)
    H.upload(env, st)
    _, _, values = pol.predict_batch(env, want_values=True)
    ref = cport.lookahead_reward(st, pol._action_table, 0.25)
    assert np.array_equal(values.cpu().numpy(), ref)


@pytest.mark.parametrize("speeds,rotations", [(3, 8), (7, 12), (1, 1)])
def test_other_action_space_sizes(speeds, rotations):
    """policy.config [action_space] speed_samples / rotation_samples (cadrl.py:82-102): the table has
    speeds * rotations + 1 rows -- 25, 85 and 2 here instead of the shipped 81 -- and the kernels take A as an argument
    ((env, action) pairs are tiled 16 at a time whatever A is)."""
    import torch
    rng = np.random.RandomState(speeds * 100 + rotations)
    E, N = 21, 5
    pol = _policy(seed=6)
    pol.speed_samples, pol.rotation_samples = speeds, rotations
    pol.action_space = None
    env = H.make_vec_env(E, N)
    st = H.random_state(rng, E, N, randomize=True)
    H.upload(env, st)
    actions, best, values = pol.predict_batch(env, want_values=True)
    values, best, actions = values.cpu().numpy(), best.cpu().numpy(), actions.cpu().numpy()
    table = pol._action_table
    A = speeds * rotations + 1
    assert table.shape == (A, 2) and values.shape == (E, A)
    w = {k: v.detach().cpu() for k, v in pol.model.state_dict().items()}
    for e in range(0, E, 3):
        row = [st.rpx[e], st.rpy[e], st.rvx[e], st.rvy[e], st.rr[e], st.rgx[e], st.rgy[e], 1.0, 0.0]
        hum = np.stack([st.hpx[e], st.hpy[e], st.hvx[e], st.hvy[e], st.hr[e]], 1)
        if np.hypot(row[5] - row[0], row[6] - row[1]) < row[4]:
            assert best[e] == -1 and tuple(actions[e]) == (0.0, 0.0)
            continue
        ref, idx = pyref.sarl_predict(w, row, hum, table)
        np.testing.assert_allclose(values[e], ref, rtol=0, atol=TOL)
        top2 = np.sort(ref)[-2:]
        if A == 2 or top2[1] - top2[0] > 2 * TOL:
            assert best[e] == idx and tuple(actions[e]) == tuple(table[idx])


def test_unicycle_lookahead_matches_oracle():
    """(v, r) action table, heading-dependent propagate and the theta feature (cadrl.py:97-99,118-124,236-237)."""
    import torch
    rng = np.random.RandomState(21)
    E, N = 24, 5
    pol = _policy(seed=5)
    pol.kinematics = "unicycle"
    pol.action_space = None
    env = H.make_vec_env(E, N, kinematics="unicycle")
    st = H.random_state(rng, E, N)
    st.rtheta[:] = rng.uniform(-3, 3, E)
    H.upload(env, st)
    actions, best, values = pol.predict_batch(env, want_values=True)
    values = values.cpu().numpy()
    table = pol._action_table
    assert table.shape == (81, 2) and table[1, 1] == -np.pi / 4          # (speed, rotation) rows
    w = {k: v.detach().cpu() for k, v in pol.model.state_dict().items()}
    for e in range(0, E, 4):
        row = [st.rpx[e], st.rpy[e], st.rvx[e], st.rvy[e], st.rr[e], st.rgx[e], st.rgy[e], 1.0, st.rtheta[e]]
        hum = np.stack([st.hpx[e], st.hpy[e], st.hvx[e], st.hvy[e], st.hr[e]], 1)
        ref, idx = pyref.sarl_predict(w, row, hum, table, kinematics="unicycle")
        np.testing.assert_allclose(values[e], ref, rtol=0, atol=TOL)


@pytest.mark.parametrize("N", [5, 10, 2])
def test_unicycle_predict_matches_reference_fixture(N, golden_dir):
    """g17_sarl_unicycle.npz = the REAL reference's predict with a unicycle robot: the 40 recorded states go through
    one predict_batch (mcn_sarl_predict, kinematics 1) -- 81 action values to the path's tolerance, the reference's
    (v, r) action wherever its top two values are further apart than that, (0, 0) where the robot stands on its goal."""
    import torch
    g = np.load(os.path.join(golden_dir, "g17_sarl_unicycle.npz"))
    pol = _policy(_weights(g, "w__"))
    pol.kinematics = "unicycle"
    pol.action_space = None
    me, hum = g["N%d_self" % N], g["N%d_humans" % N]
    E = me.shape[0]
    env = H.make_vec_env(E, N, kinematics="unicycle")
    st = H.random_state(np.random.RandomState(0), E, N)
    st.rpx[:], st.rpy[:], st.rvx[:], st.rvy[:], st.rr[:] = me[:, 0], me[:, 1], me[:, 2], me[:, 3], me[:, 4]
    st.rgx[:], st.rgy[:], st.rtheta[:] = me[:, 5], me[:, 6], me[:, 8]
    st.hpx[:], st.hpy[:], st.hvx[:], st.hvy[:], st.hr[:] = (hum[:, :, c] for c in range(5))
    H.upload(env, st)
    actions, best, values = pol.predict_batch(env, want_values=True)
    values, actions = values.cpu().numpy(), actions.cpu().numpy()
    assert np.array_equal(pol._action_table, g["table"])
    checked = 0
    for e in range(E):
        want = g["N%d_values" % N][e]
        if np.isnan(want[0]):
            assert tuple(actions[e]) == (0.0, 0.0)
            continue
        np.testing.assert_allclose(values[e], want, rtol=0, atol=TOL)
        top2 = np.sort(want)[-2:]
        if top2[1] - top2[0] > 2 * TOL:
            assert tuple(actions[e]) == tuple(g["N%d_action" % N][e]), e
            checked += 1
    assert checked > 25
    # the E = 1 surface on the same states: get_attention_weights() is what the reference's model holds after predict()
    # -- the weights of the last candidate action (sarl.py:56,88-89), unchanged where predict() returned early
    from modelcrowdnav_amd.envs.utils.state import FullState, ObservableState, JointState
    prev = None
    for e in range(0, E, 2):
        js = JointState(FullState(*me[e].tolist()), [ObservableState(*row) for row in hum[e].tolist()])
        act = pol.predict(js)
        want = g["N%d_attention" % N][e]
        if np.isnan(g["N%d_values" % N][e][0]):
            assert tuple(act) == (0, 0)
            assert prev is None or np.array_equal(pol.get_attention_weights(), prev)          # stale, as in the reference
            continue
        prev = pol.get_attention_weights()
        np.testing.assert_allclose(prev, want, rtol=0, atol=TOL)
        assert pol.chosen_attention_weights.shape == (N,) and abs(pol.chosen_attention_weights.sum() - 1) < 1e-5


def test_e1_epsilon_greedy_follows_numpys_stream_like_the_reference(golden_dir):
    """g21_epsilon.npz = the REAL reference's predict in the train phase with epsilon 0.5 under np.random.seed(2100):
    one uniform draw per call decides exploration, np.random.choice picks the table row (multi_human_rl.py:26-29), no
    draw where the robot stands on its goal (:22-23), `last_state` = transform(state) kept in the train phase only
    (:60-61).  The E = 1 predict() consumes numpy's global stream the same way: same explored calls, same random
    rows, same greedy actions, same stored states."""
    import torch
    from modelcrowdnav_amd.envs.utils.state import FullState, ObservableState, JointState
    g = np.load(os.path.join(golden_dir, "g21_epsilon.npz"))
    pol = _policy(_weights(np.load(os.path.join(golden_dir, "g17_sarl_unicycle.npz")), "w__"))
    pol.set_phase("train")
    with pytest.raises(AttributeError):
        pol.predict(JointState(FullState(*g["selfs"][0].tolist()), [ObservableState(*r) for r in g["humans"][0].tolist()]))
    pol.set_epsilon(0.5)
    np.random.seed(2100)
    kinds = np.bincount(g["explored"], minlength=3)
    assert kinds.min() >= 5
    for s in range(g["selfs"].shape[0]):
        js = JointState(FullState(*g["selfs"][s].tolist()), [ObservableState(*r) for r in g["humans"][s].tolist()])
        pol.action_values = None
        act = pol.predict(js)
        kind = int(g["explored"][s])
        assert (2 if pol.reach_destination(js) else int(pol.action_values is None)) == kind, s
        if kind == 0:                 # greedy: compare unless the device's float32 values tie differently
            vals = np.sort(np.array(pol.action_values))[-2:]
            if vals[1] - vals[0] > 2 * TOL:
                assert (act.vx, act.vy) == tuple(g["actions"][s]), s
        else:
            assert (act.vx, act.vy) == tuple(g["actions"][s]), s
        if s > 0 or kind != 2:
            np.testing.assert_allclose(pol.last_state.cpu().numpy(), g["last_states"][s], rtol=2e-6, atol=2e-6)
    pol.set_phase("test")
    pol.last_state = None
    pol.predict(JointState(FullState(*g["selfs"][0].tolist()), [ObservableState(*r) for r in g["humans"][0].tolist()]))
    assert pol.last_state is None                       # only the train phase keeps it


def test_per_env_pedestrian_counts():
    """mcn_env_state.hcount: env e shows only its first hcount[e] pedestrians to the policy.  Values must equal
    the reference network evaluated on the shorter list (attention sum, mean and distance test all ignore the
    absent slots), whatever sits in those slots."""
    import torch
    rng = np.random.RandomState(17)
    E, N = 41, 7
    pol = _policy(seed=4)
    env = H.make_vec_env(E, N)
    st = H.random_state(rng, E, N, randomize=True)
    H.upload(env, st)
    counts = rng.randint(1, N + 1, E).astype(np.int32)
    counts[0], counts[1] = N, 1
    hc = torch.from_numpy(counts).to(env.device)
    actions, best, values = pol.predict_batch(env, want_values=True, hcount=hc)
    values, best = values.cpu().numpy().copy(), best.cpu().numpy().copy()
    full = pol.predict_batch(env, want_values=True)[2].cpu().numpy()
    assert np.array_equal(values[0], full[0])                         # hcount == N is the unmasked path
    w = {k: v.detach().cpu() for k, v in pol.model.state_dict().items()}
    table = pol._action_table
    for e in range(0, E, 2):
        n = int(counts[e])
        row = [st.rpx[e], st.rpy[e], st.rvx[e], st.rvy[e], st.rr[e], st.rgx[e], st.rgy[e], 1.0, 0.0]
        hum = np.stack([st.hpx[e], st.hpy[e], st.hvx[e], st.hvy[e], st.hr[e]], 1)[:n]
        ref, idx = pyref.sarl_predict(w, row, hum, table)
        np.testing.assert_allclose(values[e], ref, rtol=0, atol=TOL)
    # garbage in the absent slots must not leak in
    env.hpos[:, 3:] = 1e6; env.hvel[:, 3:] = -7.0
    hc3 = torch.full((E,), 3, dtype=torch.int32, device=env.device)
    v3 = pol.predict_batch(env, want_values=True, hcount=hc3)[2].cpu().numpy()
    e = 5
    row = [st.rpx[e], st.rpy[e], st.rvx[e], st.rvy[e], st.rr[e], st.rgx[e], st.rgy[e], 1.0, 0.0]
    hum = np.stack([st.hpx[e], st.hpy[e], st.hvx[e], st.hvy[e], st.hr[e]], 1)[:3]
    np.testing.assert_allclose(v3[e], pyref.sarl_predict(w, row, hum, table)[0], rtol=0, atol=TOL)
    with pytest.raises(ValueError):
        pol.predict_batch(env, hcount=hc.long())


@pytest.mark.parametrize("N", [5, 10, 3])
def test_bf16x3_layers_agree_with_the_float32_mfma_layers(N):
    """The look-ahead's layers run on the bf16 matrix pipe with every float32 operand split into three bfloat16 pieces
    (mcn_sarl_net.x3, csrc/mfma_chain.hpp: dense_flow_x3); mcn_tuning.sarl_x3 = 0 keeps the float32 MFMA layers.  Both
    are float32-accurate evaluations of the same network: on the same 1 024-env batch they agree to a few float32
    rounding steps of the values (|v| ~ 1: 3e-6), far inside the 1e-5 parity bar each holds against the reference, and
    the rewards -- float64, computed outside the network -- are identical."""
    import torch
    from modelcrowdnav_amd import _hip
    rng = np.random.RandomState(70 + N)
    E = 1024
    pol = _policy(seed=5)
    env = H.make_vec_env(E, N)
    H.upload(env, H.random_state(rng, E, N, randomize=True))
    with _hip.tuned(sarl_x3=1):
        _, best_x3, v_x3 = pol.predict_batch(env, want_values=True)
        v_x3, best_x3 = v_x3.clone(), best_x3.clone()
    with _hip.tuned(sarl_x3=0):
        _, best_f32, v_f32 = pol.predict_batch(env, want_values=True)
        v_f32, best_f32 = v_f32.clone(), best_f32.clone()
    torch.cuda.synchronize()
    d = (v_x3 - v_f32).abs().max().item()
    assert d <= 3e-6, d
    assert d > 0.0, "the two paths should not be the same kernel"
    same = (best_x3 == best_f32)
    # a different argmax needs two candidate values closer than the two paths' own difference
    top2 = torch.topk(v_f32, 2, dim=1).values
    assert bool((same | ((top2[:, 0] - top2[:, 1]) < 2 * d)).all())
    assert same.float().mean().item() > 0.99


@pytest.mark.parametrize("weights", ["seed 9", "g17"])
def test_network_error_against_a_float64_evaluation(weights, golden_dir):
    """How accurate is "float32-accurate"?  The value network evaluated in float64 on the same float32 inputs and
    weights is the yardstick: the torch float32 evaluation (what the reference runs), the float32 MFMA layers and the
    bf16x3 layers are three different roundings of it.  The bf16x3 path -- six of nine piece products, float32
    accumulation -- must not be further from the float64 result than twice the float32 reference's own distance
    (+ 5e-7); two weight sets (this package's default initialisation, the reference-side network recorded in g17)."""
    import torch
    from modelcrowdnav_amd import _hip
    N, E = 5, 256
    rng = np.random.RandomState(11)
    if weights == "g17":
        g = np.load(os.path.join(golden_dir, "g17_sarl_unicycle.npz"))
        pol = _policy(weights=_weights(g, "w__"))
    else:
        pol = _policy(seed=9)
    env = H.make_vec_env(E, N)
    st = H.random_state(rng, E, N, randomize=True)
    H.upload(env, st)
    with _hip.tuned(sarl_x3=1):
        _, _, v_x3 = pol.predict_batch(env, want_values=True)
        v_x3 = v_x3.cpu().numpy().copy()
    with _hip.tuned(sarl_x3=0):
        _, _, v_f32 = pol.predict_batch(env, want_values=True)
        v_f32 = v_f32.cpu().numpy().copy()
    w32 = {k: v.detach().cpu() for k, v in pol.model.state_dict().items()}
    w64 = {k: v.double() for k, v in w32.items()}
    table, dt = pol._action_table, 0.25
    err = dict(x3=0.0, f32_mfma=0.0, torch_f32=0.0)
    for e in range(0, E, 8):
        reached = float(np.linalg.norm((st.rpy[e] - st.rgy[e], st.rpx[e] - st.rgx[e]))) < st.rr[e]
        if reached:
            continue
        nxt = np.stack([st.hpx[e] + st.hvx[e] * dt, st.hpy[e] + st.hvy[e] * dt, st.hvx[e], st.hvy[e], st.hr[e]], 1)
        rows = np.zeros((len(table), N, 14))
        rows[:, :, 0] = (st.rpx[e] + table[:, 0] * dt)[:, None]; rows[:, :, 1] = (st.rpy[e] + table[:, 1] * dt)[:, None]
        rows[:, :, 2] = table[:, 0][:, None]; rows[:, :, 3] = table[:, 1][:, None]
        rows[:, :, 4] = st.rr[e]; rows[:, :, 5] = st.rgx[e]; rows[:, :, 6] = st.rgy[e]; rows[:, :, 7] = 1.0
        rows[:, :, 9:] = nxt[None]
        feats = pyref.rotate(torch.from_numpy(rows.reshape(-1, 14)).float()).view(len(table), N, 13)
        truth = pyref.sarl_forward(w64, feats.double())[0].numpy()
        ref32 = pyref.sarl_forward(w32, feats)[0].double().numpy()
        rew = np.array([pyref.lookahead_reward(rows[a, 0, 0], rows[a, 0, 1], st.rr[e], st.rgx[e], st.rgy[e],
                                               [(h[0], h[1], h[4]) for h in nxt], dt) for a in range(len(table))])
        disc = 0.9 ** (dt * 1.0)
        err["x3"] = max(err["x3"], float(np.abs((v_x3[e] - rew) / disc - truth).max()))
        err["f32_mfma"] = max(err["f32_mfma"], float(np.abs((v_f32[e] - rew) / disc - truth).max()))
        err["torch_f32"] = max(err["torch_f32"], float(np.abs(ref32 - truth).max()))
    print("SARL network error vs float64 (weights: %s): %s" % (weights, ", ".join("%s %.2e" % kv for kv in err.items())))
    assert err["torch_f32"] > 0
    assert err["x3"] <= 2 * err["torch_f32"] + 5e-7, err
    assert err["f32_mfma"] <= 2 * err["torch_f32"] + 5e-7, err


@pytest.mark.parametrize("N", [5, 10])
def test_benchmark_grid_every_pair_against_the_torch_reference(N):
    """BASELINE configs 3 / 4 in full: all 4096 x 81 (env, action) values of one predict_batch against the torch-fp32
    restatement of MultiHumanRL.predict (multi_human_rl.py:35-63: propagate, reward ladder, rotate, ValueNetwork.forward),
    evaluated batched on the CPU -- not a sample of envs.  Bar 1e-5 on every value; the chosen action is the reference's
    wherever its top two values are further apart than that."""
    import torch
    rng = np.random.RandomState(90 + N)
    E, dt, gamma = 4096, 0.25, 0.9
    pol = _policy(seed=4)
    env = H.make_vec_env(E, N)
    st = H.random_state(rng, E, N, randomize=True)
    H.upload(env, st)
    actions, best, values = pol.predict_batch(env, want_values=True)
    torch.cuda.synchronize()
    values, best = values.cpu().numpy(), best.cpu().numpy()
    table = np.asarray(pol._action_table, np.float64)
    A = len(table)
    w = {k: v.detach().cpu() for k, v in pol.model.state_dict().items()}
    # propagate (float64), one row per (env, action, human)
    npx = st.rpx[:, None] + table[None, :, 0] * dt; npy = st.rpy[:, None] + table[None, :, 1] * dt          # [E, A]
    hx, hy = st.hpx + st.hvx * dt, st.hpy + st.hvy * dt                                                      # [E, N]
    rows = np.empty((E, A, N, 14))
    rows[..., 0] = npx[:, :, None]; rows[..., 1] = npy[:, :, None]
    rows[..., 2] = table[None, :, None, 0]; rows[..., 3] = table[None, :, None, 1]
    rows[..., 4] = st.rr[:, None, None]; rows[..., 5] = st.rgx[:, None, None]; rows[..., 6] = st.rgy[:, None, None]
    rows[..., 7] = 1.0; rows[..., 8] = 0.0
    rows[..., 9] = hx[:, None, :]; rows[..., 10] = hy[:, None, :]
    rows[..., 11] = st.hvx[:, None, :]; rows[..., 12] = st.hvy[:, None, :]; rows[..., 13] = st.hr[:, None, :]
    # reward ladder (multi_human_rl.py:65-88), float64
    d = np.sqrt((npx[:, :, None] - hx[:, None, :]) ** 2 + (npy[:, :, None] - hy[:, None, :]) ** 2) \
        - st.rr[:, None, None] - st.hr[:, None, :]                                                          # [E, A, N]
    coll = (d < 0).any(2)
    dmin = d.min(2)
    reach = np.sqrt((npx - st.rgx[:, None]) ** 2 + (npy - st.rgy[:, None]) ** 2) < st.rr[:, None]
    rew = np.where(coll, -0.25, np.where(reach, 1.0, np.where(dmin < 0.2, (dmin - 0.2) * 0.5 * dt, 0.0)))
    on_edge = (np.abs(d) < 1e-9).any(2) | (np.abs(dmin - 0.2) < 1e-9)        # a rung decided by the last bit: not compared
    net = np.empty((E, A))
    flat = torch.from_numpy(rows.reshape(E * A, N, 14))
    with torch.no_grad():
        for lo in range(0, E * A, 1 << 15):
            chunk = flat[lo:lo + (1 << 15)].float()
            B = chunk.shape[0]
            feats = pyref.rotate(chunk.reshape(B * N, 14)).view(B, N, 13)
            net.reshape(-1)[lo:lo + B] = pyref.sarl_forward(w, feats)[0].double().numpy()
    ref = rew + gamma ** (dt * 1.0) * net
    reached = np.sqrt((st.rpx - st.rgx) ** 2 + (st.rpy - st.rgy) ** 2) < st.rr       # policy.py:43-49: nothing evaluated
    ok = ~reached[:, None] & ~on_edge
    err = np.abs(values - ref)[ok]
    print("SARL 4096 x 81 x %d: %d of %d values compared, max |value - reference| = %.3g" % (N, ok.sum(), E * A, err.max()))
    assert ok.mean() > 0.9            # (random_state puts ~6 % of the robots on their goal: nothing is evaluated there)
    assert err.max() <= TOL
    moving = ~reached & ~on_edge.any(1)
    top2 = np.sort(ref, 1)[:, -2:]
    clear = moving & (top2[:, 1] - top2[:, 0] > 2 * TOL)
    assert clear.sum() > E // 2
    assert np.array_equal(best[clear], ref[clear].argmax(1))
    assert np.all(best[reached] == -1)
