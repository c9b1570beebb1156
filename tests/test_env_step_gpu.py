"""GPU parity of mcn_env_step (through the C ABI) against the C oracle and the reference fixtures.

Bar (BASELINE.json north_star): done / info / collision masks and integer counts bit-exact,
float state within 1e-5.  Achieved here: float state bit-exact for ORCA / given-velocity
humans; `linear` humans use device atan2/cos/sin and are held to 1e-12.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import cport  # noqa: E402
from tests import helpers as H  # noqa: E402


def _torch():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _step_both(env, st, ax, ay, update, policy=cport.HUMANS_ORCA, given=None):
    torch = _torch()
    H.upload(env, st)
    act = torch.from_numpy(np.stack([ax, ay], -1)).to(env.device)
    gv = None if given is None else torch.from_numpy(np.ascontiguousarray(given)).to(env.device)
    if policy == cport.HUMANS_LINEAR:
        env.human_policy_name = "linear"
    ob, reward, done, info = env.step(act, update=update, given_v=gv)
    torch.cuda.synchronize()
    ref_st = st.copy()
    ref = cport.env_step(H.oracle_cfg_for(env, policy), ref_st, ax, ay, update=update, given_v=given)
    got = dict(reward=reward.cpu().numpy(), done=done.cpu().numpy(), info=info.cpu().numpy(),
               dmin=env.dmin.cpu().numpy(), hh_count=env.hh_count.cpu().numpy(),
               human_act=env.human_act.cpu().numpy())
    if not update:
        got.update(nobs_px=ob.pos[..., 0].cpu().numpy(), nobs_py=ob.pos[..., 1].cpu().numpy(),
                   nobs_vx=ob.vel[..., 0].cpu().numpy(), nobs_vy=ob.vel[..., 1].cpu().numpy())
    return got, ref, ref_st


@pytest.mark.parametrize("N,visible,randomize", [(5, False, False), (5, True, False), (10, False, True),
                                                 (10, True, True), (1, False, False), (2, True, False),
                                                 (7, False, True), (9, True, False), (13, False, True),
                                                 (32, False, False)])
@pytest.mark.parametrize("update", [True, False])
@pytest.mark.parametrize("kernel", ["auto", "lane-per-human", "lane-per-human-256", "deferred-lp3", "deferred-lp3-256",
                                    "run-time-N", "run-time-N-256", "quad", "quad-split"])
def test_step_matches_oracle_bitexact(N, visible, randomize, update, kernel, tuning):
    # the same arithmetic exists in several decompositions (env_step.hip with compile-time or run-time N,
    # env_step_quad.hip +- wavefront split); mcn_set_tuning (the `tuning` fixture) pins which one the dispatcher picks
    if kernel.startswith("lane-per-human"):
        tuning(quad_max_envs=0)
        tuning(lp3_defer=0)
    elif kernel.startswith("deferred-lp3"):
        # the 3-D LPs parked in mcn_env_out.lp3_queue and finished by env_lp3_kernel, one per lane (what large
        # batches get), forced at a small size
        if N > 10:
            pytest.skip("the deferred path covers the compile-time-N kernels (<= 10 humans)")
        tuning(quad_max_envs=0)
        tuning(lp3_defer=1)
    elif kernel.startswith("run-time-N"):
        tuning(quad_max_envs=0)
        tuning(force_generic=1)
    if kernel.endswith("-256"):
        # the 4-wavefront workgroups that batches above 4096 wavefronts get (pair rows parked across wavefronts,
        # the register-resident 3-D LP `lp3_static` instead of the wavefront-cooperative one): forced at a small size
        tuning(step_block=256)
    elif kernel.startswith("quad"):
        if N - 1 + int(visible) > 4:
            pytest.skip("quad kernel handles at most 4 neighbours")
        tuning(quad_max_envs=1 << 30)
        tuning(quad_split=1 if kernel == "quad-split" else 0)
    rng = np.random.RandomState(100 + N + 7 * visible)
    E = 777   # ragged: not a multiple of the envs-per-wave count
    env = H.make_vec_env(E, N, robot_visible=visible)
    st = H.random_state(rng, E, N, randomize=randomize)
    sp, aa = rng.uniform(0, 1, E), rng.uniform(0, 2 * np.pi, E)
    cport.lp3_entries(reset=True)
    got, ref, ref_st = _step_both(env, st, sp * np.cos(aa), sp * np.sin(aa), update)
    # the inputs reach the rare paths: humans that fall through to the 3-D LP, overlapping human pairs
    if N >= 5:
        assert cport.lp3_entries() > 20
        assert int(ref["hh_count"].sum()) > 20
    for k in ("done", "info", "hh_count"):
        assert np.array_equal(got[k], ref[k]), k
    for k in [k for k in ref if k not in ("done", "info", "hh_count")]:
        assert np.array_equal(got[k], ref[k]), k
    if update:
        H.assert_state_equal(H.download(env), ref_st, what="N=%d" % N)
    else:
        H.assert_state_equal(H.download(env), st, what="lookahead must not mutate")


@pytest.mark.parametrize("N,visible,block", [(10, False, 256), (10, False, 64), (6, True, 256), (8, False, 64)])
@pytest.mark.parametrize("update", [True, False])
def test_full_lp3_queue_turns_lanes_back_to_solving_in_place(N, visible, block, update, tuning):
    """The deferred-3-D-LP queue holds half of the worst case (lp3_queue.hpp).  Crowds packed into overlapping
    discs send most humans into the 3-D LP -- far more than the sub-queues hold: the lanes past a sub-queue's capacity
    finish their problem in the step kernel (register-resident or wavefront-cooperative, by workgroup shape), the rest
    are parked, and every output is still the oracle's, bit for bit."""
    from modelcrowdnav_amd import _hip
    tuning(quad_max_envs=0)
    tuning(lp3_defer=1)
    tuning(step_block=block)
    rng = np.random.RandomState(31 + N)
    E = 8192 + 37
    env = H.make_vec_env(E, N, robot_visible=visible)
    st = H.random_state(rng, E, N, randomize=True)
    # every env's humans inside a disc of radius 0.45 around one point: every pair overlaps
    cx, cy = rng.uniform(-3, 3, (E, 1)), rng.uniform(-3, 3, (E, 1))
    ang, d = rng.uniform(0, 2 * np.pi, (E, N)), rng.uniform(0.0, 0.45, (E, N))
    st.hpx[:], st.hpy[:] = cx + d * np.cos(ang), cy + d * np.sin(ang)
    sp, aa = rng.uniform(0, 1, E), rng.uniform(0, 2 * np.pi, E)
    cport.lp3_entries(reset=True)
    got, ref, ref_st = _step_both(env, st, sp * np.cos(aa), sp * np.sin(aa), update)
    waves = (E + 64 // N - 1) // (64 // N) + 3
    subcap = max(64, ((waves + 255) // 256 * 64 // 2 + 63) // 64 * 64)      # lp3_queue.hpp: lp3_subcap
    assert int(_hip.lib.mcn_env_lp3_queue_bytes(E, N)) < 256 * subcap * (28 + 16 * N) + (1 << 17)
    assert cport.lp3_entries() > 1.2 * 256 * subcap, (cport.lp3_entries(), 256 * subcap)
    for k in ref:
        assert np.array_equal(got[k], ref[k]), k
    if update:
        H.assert_state_equal(H.download(env), ref_st, what="N=%d" % N)
    # the queue's counters were reset by the finish kernel: a second step from the same state gives the same bits
    got2, _, _ = _step_both(env, st, sp * np.cos(aa), sp * np.sin(aa), update)
    for k in ref:
        assert np.array_equal(got2[k], ref[k]), k


@pytest.mark.parametrize("E,N,mode", [(1 << 20, 5, "orca"), (1 << 22, 5, "orca"), (1 << 22, 5, "given"),
                                      (1 << 20, 5, "given"), (1 << 18, 10, "orca")])
def test_roofline_sweep_sizes_every_env_against_the_oracle(E, N, mode):
    """The batches bench.py's `roofline_sweep` times (2^20 and 2^22 envs of 5 humans through the fused ORCA kernel and the
    streaming pairwise kernel with non-temporal streams; 2^18 envs of 10 humans, where the dispatcher parks the 3-D LPs
    in the bounded queue by itself): one step of EVERY env against the C oracle, every byte of state and outputs -- the
    kernels that are measured at these sizes are the kernels that are right at these sizes (64-bit offsets, the
    XCD-chunked grid, the last partial workgroup)."""
    torch = _torch()
    from modelcrowdnav_amd import _hip
    rng = np.random.RandomState(E % 1000 + N)
    E = E - 3                      # ragged against every tile size
    env = H.make_vec_env(E, N)
    st = H.random_state(rng, E, N, randomize=True)
    sp, aa = rng.uniform(0, 1, E), rng.uniform(0, 2 * np.pi, E)
    if mode == "given":
        env.count_hh = False       # ModelCrowdSim.step: no human-human count (the streaming kernel's contract)
        env.track_human_times = False
        env.export_human_actions = False
        gv = rng.uniform(-1, 1, (E, N, 2))
        got, ref, ref_st = _step_both(env, st, sp * np.cos(aa), sp * np.sin(aa), True, cport.HUMANS_GIVEN, gv)
        assert _hip.last_dispatch() == "env_pair_kernel"
    else:
        got, ref, ref_st = _step_both(env, st, sp * np.cos(aa), sp * np.sin(aa), True)
        assert _hip.last_dispatch() == "env_step_kernel"
    for k in ref:
        if k == "human_act" and mode == "given":
            continue
        assert np.array_equal(got[k], ref[k]), k
    fields = [f for f in H.STATE_FIELDS if not (mode == "given" and f == "human_times")]
    H.assert_state_equal(H.download(env), ref_st, fields=fields, what="E=%d N=%d %s" % (E, N, mode))
    del env
    torch.cuda.empty_cache()


def test_given_velocity_and_linear_modes():
    rng = np.random.RandomState(5)
    E, N = 1000, 5
    env = H.make_vec_env(E, N)
    env.count_hh = False
    st = H.random_state(rng, E, N)
    gv = rng.uniform(-1, 1, (E, N, 2))
    ax, ay = rng.uniform(-1, 1, E), rng.uniform(-1, 1, E)
    got, ref, ref_st = _step_both(env, st, ax, ay, True, cport.HUMANS_GIVEN, gv)
    for k in ref:
        assert np.array_equal(got[k], ref[k]), k
    H.assert_state_equal(H.download(env), ref_st)
    env2 = H.make_vec_env(E, N)
    got, ref, ref_st = _step_both(env2, st, ax, ay, True, cport.HUMANS_LINEAR)
    assert np.array_equal(got["done"], ref["done"]) and np.array_equal(got["info"], ref["info"])
    np.testing.assert_allclose(got["human_act"], ref["human_act"], rtol=0, atol=1e-12)
    d = H.download(env2)
    np.testing.assert_allclose(d.hpx, ref_st.hpx, rtol=0, atol=1e-12)


@pytest.mark.parametrize("N", [5, 10])
@pytest.mark.parametrize("book", ["pool", "record", "none"])
@pytest.mark.parametrize("stream", [1, 2])      # 2: the non-temporal form large batches get, forced at a small size
def test_streaming_pair_kernel_equals_lane_per_human(N, book, stream, tuning):
    """env_pair.hip (given velocities, per-env data loaded / stored in 16-byte pieces by all lanes of the env, ladder
    and Explorer record on every lane) against env_step_kernel<GIVEN>: every byte of state, step record, Explorer
    record and finished-episode records over 130 steps with collisions, goals, timeouts and pool restarts; the first
    step also against the C oracle."""
    torch = _torch()
    E, T = 1003, 130
    rng = np.random.RandomState(40 + N)

    def make():
        if book == "none":
            env = H.make_vec_env(E, N)
            from modelcrowdnav_amd.envs import scenarios as S
            env.load_scenarios(S.scenario_pool(env.spec(), "test", range(64), N, "circle_crossing")[np.arange(E) % 64])
        else:
            env = _rollout_env(E, N, False, book == "pool", fin_slots=2)
        env.count_hh = False
        return env
    a, b = make(), make()
    # robots head for their goal (0, 4) with some noise; humans drift across the circle
    ang = rng.uniform(0, 2 * np.pi, (T, E, N))
    gv = torch.from_numpy(np.stack([0.6 * np.cos(ang), 0.6 * np.sin(ang)], -1)).to(a.device)
    acts = torch.from_numpy(np.stack([rng.uniform(-0.3, 0.3, (T, E)), rng.uniform(0.2, 1.0, (T, E))], -1)).to(a.device)
    st0 = H.download(a)
    for t in range(T):
        tuning(pair_stream=stream)
        a.step(acts[t], given_v=gv[t])
        tuning(pair_stream=0)
        b.step(acts[t], given_v=gv[t])
        if t == 0:
            ref = cport.env_step(H.oracle_cfg_for(a, cport.HUMANS_GIVEN), st0, acts[0, :, 0].cpu().numpy().copy(),
                                 acts[0, :, 1].cpu().numpy().copy(), update=True, given_v=gv[0].cpu().numpy())
            assert np.array_equal(a.done.cpu().numpy(), ref["done"]) and np.array_equal(a.reward.cpu().numpy(), ref["reward"])
            assert np.array_equal(a.dmin.cpu().numpy(), ref["dmin"]) and np.array_equal(a.info.cpu().numpy(), ref["info"])
            if book != "pool":
                H.assert_state_equal(H.download(a), st0, what="streaming pair kernel vs oracle")
        if t % 13 == 0 or t == T - 1:
            torch.cuda.synchronize()
            sa, sb = _snapshot(a) if book != "none" else None, _snapshot(b) if book != "none" else None
            if book == "none":
                for k in ("hpos", "hvel", "rpos", "rvel", "gtime", "step_rec"):
                    assert np.array_equal(getattr(a, k).cpu().numpy().view(np.uint8), getattr(b, k).cpu().numpy().view(np.uint8)), (k, t)
            else:
                for k in sa:
                    assert np.array_equal(sa[k].view(np.uint8), sb[k].view(np.uint8)), (k, t)
    if book != "none":
        assert int(a.rollout_buffers["fin_count"].sum().item()) > E // 2


@pytest.mark.parametrize("name,policy", [("g2_step_given", cport.HUMANS_GIVEN), ("g2_step_linear", cport.HUMANS_LINEAR),
                                         ("g2_step_orca", cport.HUMANS_ORCA),
                                         ("g2_step_orca_visible", cport.HUMANS_ORCA)])
def test_step_matches_reference_fixtures(name, policy, golden_dir):
    """HIP path against values recorded from the real reference env (tests/golden_tools/gen_golden.py)."""
    torch = _torch()
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    visible = bool(g["robot_visible"])
    tol = 1e-12 if policy == cport.HUMANS_LINEAR else 0.0
    for N in np.unique(g["N"]):
        for update in (True, False):
            sel = np.where((g["N"] == N) & (g["update"] == update))[0]
            if len(sel) == 0:
                continue
            E, N = len(sel), int(N)
            env = H.make_vec_env(E, N, robot_visible=visible)
            env.count_hh = policy != cport.HUMANS_GIVEN
            st = cport.EnvState(E, N)
            h, r = g["hum_in"][sel][:, :N], g["rob_in"][sel]
            st.hpx[:], st.hpy[:], st.hvx[:], st.hvy[:], st.hr[:] = h[..., 0], h[..., 1], h[..., 2], h[..., 3], h[..., 4]
            st.hgx[:], st.hgy[:], st.hvpref[:] = h[..., 5], h[..., 6], h[..., 7]
            st.rpx[:], st.rpy[:], st.rvx[:], st.rvy[:], st.rr[:] = r[:, 0], r[:, 1], r[:, 2], r[:, 3], r[:, 4]
            st.rgx[:], st.rgy[:] = r[:, 5], r[:, 6]
            st.gtime[:] = g["time"][sel]
            H.upload(env, st)
            env.human_policy_name = {cport.HUMANS_LINEAR: "linear"}.get(policy, "orca")
            act = torch.from_numpy(g["act"][sel]).to(env.device)
            gv = torch.from_numpy(g["given_v"][sel][:, :N].copy()).to(env.device) if policy == cport.HUMANS_GIVEN else None
            ob, reward, done, info = env.step(act, update=bool(update), given_v=gv)
            assert np.array_equal(done.cpu().numpy(), g["done"][sel].astype(np.uint8))
            assert np.array_equal(info.cpu().numpy(), g["info"][sel].astype(np.uint8))
            np.testing.assert_allclose(reward.cpu().numpy(), g["reward"][sel], rtol=0, atol=tol)
            danger = g["info"][sel] == 1
            np.testing.assert_allclose(env.dmin.cpu().numpy()[danger], g["dmin"][sel][danger], rtol=0, atol=tol)
            want = g["obs"][sel][:, :N]
            np.testing.assert_allclose(ob.pos.cpu().numpy(), want[..., 0:2], rtol=0, atol=tol)
            np.testing.assert_allclose(ob.vel.cpu().numpy(), want[..., 2:4], rtol=0, atol=tol)
            if update:
                ro = g["rob_out"][sel]
                np.testing.assert_allclose(env.rpos.cpu().numpy(), ro[:, 0:2], rtol=0, atol=tol)
                np.testing.assert_allclose(env.rvel.cpu().numpy(), ro[:, 2:4], rtol=0, atol=tol)
                np.testing.assert_allclose(env.gtime.cpu().numpy(), g["time_out"][sel], rtol=0, atol=0)
                if policy != cport.HUMANS_GIVEN:
                    np.testing.assert_allclose(env.human_times.cpu().numpy(), g["human_times"][sel][:, :N], rtol=0, atol=0)


@pytest.mark.parametrize("kernel", ["auto", "lane-per-human"])
def test_rollout_4096x5_bitexact_trajectory(kernel, tuning):
    if kernel == "lane-per-human":
        tuning(quad_max_envs=0)
    _rollout_4096x5()


def _rollout_4096x5():
    """BASELINE config 2 shape: 4096 envs x 5 humans, ORCA humans, random robot actions, 60 steps.
    Whole trajectories must stay bit-identical to the oracle (any 1-ulp slip would compound)."""
    torch = _torch()
    from modelcrowdnav_amd.envs import scenarios as S
    E, N, T = 4096, 5, 60
    env = H.make_vec_env(E, N)
    env.reset("test", test_cases=[i % 500 for i in range(E)])
    st = H.download(env)
    cfg = H.oracle_cfg_for(env)
    rng = np.random.RandomState(0)
    done_total = 0
    for t in range(T):
        sp, aa = rng.uniform(0, 1, E), rng.uniform(0, 2 * np.pi, E)
        ax, ay = sp * np.cos(aa), sp * np.sin(aa)
        ob, reward, done, info = env.step(torch.from_numpy(np.stack([ax, ay], -1)).to(env.device))
        ref = cport.env_step(cfg, st, ax, ay, update=True)
        assert np.array_equal(done.cpu().numpy(), ref["done"]), t
        assert np.array_equal(info.cpu().numpy(), ref["info"]), t
        assert np.array_equal(reward.cpu().numpy(), ref["reward"]), t
        assert np.array_equal(env.hh_count.cpu().numpy(), ref["hh_count"]), t
        done_total += int(ref["done"].sum())
    H.assert_state_equal(H.download(env), st, what="after %d steps" % T)
    assert done_total > 0


def _rollout_env(E, N, visible, with_pool, kinematics="holonomic", fin_slots=1):
    from modelcrowdnav_amd.envs import scenarios as S
    env = H.make_vec_env(E, N, robot_visible=visible, kinematics=kinematics)
    pool = S.scenario_pool(env.spec(), "test", range(64), N, "circle_crossing")
    ids = np.arange(E) % 64
    env.load_scenarios(pool[ids])
    env.attach_rollout(gamma=0.9, pool=pool if with_pool else None, case_stride=3, first_cases=(ids + 7) % 64,
                       fin_slots=fin_slots)
    return env


def _snapshot(env):
    c = lambda t: t.detach().cpu().numpy().copy()
    snap = {k: c(getattr(env, k)) for k in ("hpos", "hvel", "hgoal", "hrad", "hvpref", "rpos", "rvel", "rgoal",
                                            "rtheta", "gtime", "human_times", "step_rec", "human_act")}
    snap.update({"roll_" + k: c(v) for k, v in env.rollout_buffers.items() if k in ("state", "fin_return", "fin_time",
                                                                                  "fin_info")})
    return snap


@pytest.mark.parametrize("N,visible", [(5, False), (4, True), (3, False), (1, True), (2, False)])
@pytest.mark.parametrize("with_pool", [True, False])
@pytest.mark.parametrize("split", ["0", "1"])
def test_rollout_launch_equals_single_steps(N, visible, with_pool, split, tuning):
    """mcn_env_rollout (T steps in one launch, state in registers) == T mcn_env_step calls, every byte of state,
    step record, Explorer accounting and finished-episode records; episodes end and restart inside the sequence."""
    torch = _torch()
    E, T = 1000, 110                    # ragged vs the envs-per-wavefront tiling; every env passes the time limit
    rng = np.random.RandomState(N)
    sp, aa = rng.uniform(0, 1, (T, E)), rng.uniform(0, 2 * np.pi, (T, E))
    acts = torch.from_numpy(np.stack([sp * np.cos(aa), sp * np.sin(aa)], -1))
    tuning(rollout_fused=1)
    tuning(rollout_split=int(split))     # one wavefront per env group / two cooperating ones
    a = _rollout_env(E, N, visible, with_pool, fin_slots=2)
    b = _rollout_env(E, N, visible, with_pool, fin_slots=2)
    acts_d = acts.to(a.device)
    a.rollout(acts_d[:30]); a.rollout(acts_d[30:31]); a.rollout(acts_d[31:])          # split launches compose
    for t in range(T):
        b.step(acts_d[t])
    torch.cuda.synchronize()
    sa, sb = _snapshot(a), _snapshot(b)
    for k in sa:
        assert np.array_equal(sa[k].view(np.uint8), sb[k].view(np.uint8)), k
    assert int(a.rollout_buffers["fin_count"].min().item()) >= 1
    tuning(rollout_fused=0)                # the T-launch path of the same entry point
    c = _rollout_env(E, N, visible, with_pool, fin_slots=2)
    c.rollout(acts_d)
    torch.cuda.synchronize()
    sc = _snapshot(c)
    for k in sa:
        assert np.array_equal(sc[k].view(np.uint8), sb[k].view(np.uint8)), k


@pytest.mark.parametrize("N,visible", [(10, False), (7, True), (6, False), (5, True)])
@pytest.mark.parametrize("defer", [0, 1])
def test_rollout_entry_point_with_larger_crowds(N, visible, defer, tuning):
    """mcn_env_rollout for crowds the quad-parallel rollout kernel does not cover (more than 4 ORCA neighbours per
    human): defer = 0 takes env_step_loop_kernel -- the one-wavefront step kernel (cooperative 3-D LP included) run T
    times inside ONE launch --, defer = 1 the T single-step launches with parked 3-D LPs; both must leave every byte as T
    mcn_env_step calls do (the oracle comparison over whole trajectories is test_larger_crowd_trajectories_match_oracle
    and, for the looped launch, test_looped_rollout_launch_matches_oracle_trajectory)."""
    torch = _torch()
    E, T = 300, 110
    rng = np.random.RandomState(N)
    sp, aa = rng.uniform(0, 1, (T, E)), rng.uniform(0, 2 * np.pi, (T, E))
    acts = torch.from_numpy(np.stack([sp * np.cos(aa), sp * np.sin(aa)], -1))
    a = _rollout_env(E, N, visible, True, fin_slots=2)
    b = _rollout_env(E, N, visible, True, fin_slots=2)
    acts_d = acts.to(a.device)
    # defer = 1: a's launches park the 3-D LPs (pool restarts, first-arrival times and exported actions then come
    # from two kernels); b always solves them in the step kernel
    tuning(lp3_defer=defer)
    a.rollout(acts_d[:30]); a.rollout(acts_d[30:])
    tuning(lp3_defer=0)
    for t in range(T):
        b.step(acts_d[t])
    torch.cuda.synchronize()
    sa, sb = _snapshot(a), _snapshot(b)
    for k in sa:
        assert np.array_equal(sa[k].view(np.uint8), sb[k].view(np.uint8)), k
    assert int(a.rollout_buffers["fin_count"].min().item()) >= 1


@pytest.mark.parametrize("N,visible", [(10, False), (7, True), (9, False), (8, True), (6, False), (5, True)])
def test_looped_rollout_launch_matches_oracle_trajectory(N, visible, tuning):
    """env_step_loop_kernel (mcn_env_rollout for 6-10 ORCA humans in a latency-bound batch: T steps in ONE launch, the
    state going through L2 between iterations) against the oracle stepping the same 60-step action sequence: the state
    after the launch, bit for bit -- and the same launch with rollout_fused = 0 (T step launches) for the record."""
    torch = _torch()
    E, T = 700, 60
    rng = np.random.RandomState(100 + N)
    sp, aa = rng.uniform(0, 1, (T, E)), rng.uniform(0, 2 * np.pi, (T, E))
    ax, ay = sp * np.cos(aa), sp * np.sin(aa)
    env = H.make_vec_env(E, N, robot_visible=visible)
    env.reset("test", test_cases=[i % 500 for i in range(E)])
    st = H.download(env)
    cfg = H.oracle_cfg_for(env)
    cport.lp3_entries(reset=True)
    tuning(lp3_defer=0)
    env.rollout(torch.from_numpy(np.stack([ax, ay], -1)).to(env.device))
    for t in range(T):
        ref = cport.env_step(cfg, st, ax[t], ay[t], update=True)
    H.assert_state_equal(H.download(env), st, what="after one %d-step launch" % T)
    assert np.array_equal(env.reward.cpu().numpy(), ref["reward"]) and np.array_equal(env.done.cpu().numpy(), ref["done"])
    assert np.array_equal(env.human_act.cpu().numpy(), ref["human_act"])          # outputs of the last step
    assert cport.lp3_entries() > 100, "the crossing should drive humans into the 3-D LP"


@pytest.mark.parametrize("N", [7, 10, 5])
@pytest.mark.parametrize("block", [64, 256, "deferred"])
def test_larger_crowd_trajectories_match_oracle(N, block, tuning):
    """Crowds of 7 / 10 (test_mul_env.py:31-33, BASELINE config 5) over 40 steps of circle crossing against the
    oracle's trajectory, bit for bit, through both lane-per-human decompositions: one-wavefront workgroups with the
    wavefront-cooperative 3-D LP (what batches up to 4096 wavefronts get) and 4-wavefront workgroups with the
    register-resident one (larger batches).  The crossing is dense enough that hundreds of humans per step take the
    3-D LP."""
    torch = _torch()
    E, T = 1024, 40
    tuning(quad_max_envs=0)
    if block == "deferred":
        tuning(step_block=256)
        tuning(lp3_defer=1)            # + the second, dense launch that finishes the parked 3-D LPs
    else:
        tuning(step_block=block)
        tuning(lp3_defer=0)
    env = H.make_vec_env(E, N)
    env.reset("test", test_cases=[i % 500 for i in range(E)])
    st = H.download(env)
    cfg = H.oracle_cfg_for(env)
    rng = np.random.RandomState(N)
    cport.lp3_entries(reset=True)
    for t in range(T):
        sp, aa = rng.uniform(0, 1, E), rng.uniform(0, 2 * np.pi, E)
        ax, ay = sp * np.cos(aa), sp * np.sin(aa)
        ob, reward, done, info = env.step(torch.from_numpy(np.stack([ax, ay], -1)).to(env.device))
        ref = cport.env_step(cfg, st, ax, ay, update=True)
        assert np.array_equal(done.cpu().numpy(), ref["done"]), t
        assert np.array_equal(reward.cpu().numpy(), ref["reward"]), t
        assert np.array_equal(env.hh_count.cpu().numpy(), ref["hh_count"]), t
        assert np.array_equal(env.human_act.cpu().numpy(), ref["human_act"]), t
    H.assert_state_equal(H.download(env), st, what="after %d steps" % T)
    assert cport.lp3_entries() > (100 if N > 5 else 20) * T // 10, "the crossing should drive humans into the 3-D LP"


@pytest.mark.parametrize("split", ["0", "1"])
def test_rollout_launch_unicycle_robot(split, tuning):
    """The (v, r) robot (float64 sin / cos on the device, agent.py:110-135) through the fused launch: same code as the
    single step, so the same bytes; the trig-free holonomic kernel is a separate instantiation."""
    torch = _torch()
    E, N, T = 500, 5, 60
    rng = np.random.RandomState(8)
    acts = torch.from_numpy(np.stack([rng.uniform(0, 1, (T, E)), rng.uniform(-np.pi / 4, np.pi / 4, (T, E))], -1))
    tuning(rollout_fused=1)
    tuning(rollout_split=int(split))
    a = _rollout_env(E, N, False, True, kinematics="unicycle")
    b = _rollout_env(E, N, False, True, kinematics="unicycle")
    acts_d = acts.to(a.device)
    a.rollout(acts_d[:37]); a.rollout(acts_d[37:])
    for t in range(T):
        b.step(acts_d[t])
    torch.cuda.synchronize()
    sa, sb = _snapshot(a), _snapshot(b)
    for k in sa:
        assert np.array_equal(sa[k].view(np.uint8), sb[k].view(np.uint8)), k
    assert float(a.rtheta.abs().max()) > 0.1 and int(a.rollout_buffers["fin_count"].sum().item()) > 0


def test_rollout_bench_configuration_equals_single_steps():
    """The benchmark's own workload (bench.build_env: 4096 envs x 5 humans, 500-case pool, 81-entry action table):
    250-step mcn_env_rollout launches against one mcn_env_step launch per step, every byte of state and records."""
    torch = _torch()
    import bench
    dev = torch.device("cuda", 0)
    E, N, T = 4096, 5, 400
    acts = bench.make_actions(T, E, E, 0, dev)
    a, _ = bench.build_env(E, N, 0, dev)
    b, _ = bench.build_env(E, N, 0, dev)
    a.rollout(acts[:250]); a.rollout(acts[250:])
    for t in range(T):
        b.step(acts[t])
    torch.cuda.synchronize()
    for k in ("hpos", "hvel", "hgoal", "hrad", "hvpref", "rpos", "rvel", "rgoal", "gtime", "step_rec"):
        assert np.array_equal(getattr(a, k).cpu().numpy().view(np.uint8), getattr(b, k).cpu().numpy().view(np.uint8)), k
    for k in ("state", "fin_return", "fin_time", "fin_info"):
        assert np.array_equal(a.rollout_buffers[k].cpu().numpy().view(np.uint8),
                              b.rollout_buffers[k].cpu().numpy().view(np.uint8)), k
    assert int(a.rollout_buffers["fin_count"].sum().item()) > 10000


def _oracle_with_pool_restarts(env_ids, pool, N, acts_np, T, env_for_cfg):
    """The oracle stepping the sampled envs through the bench workload: on done the env restarts from the scenario pool
    exactly as the in-kernel reset does (mcn.h mcn_rollout: humans <- pool[next_case], zero velocity, first-arrival
    times cleared, robot back to its start pose, clock 0, next_case += case_stride mod pool_size)."""
    from modelcrowdnav_amd.envs import scenarios as S
    n = len(env_ids)
    st = cport.EnvState(n, N)

    def load(rows, cases):
        sc = pool[cases]
        st.hpx[rows], st.hpy[rows], st.hgx[rows], st.hgy[rows] = sc[..., S.PX], sc[..., S.PY], sc[..., S.GX], sc[..., S.GY]
        st.hvx[rows], st.hvy[rows] = sc[..., S.VX], sc[..., S.VY]
        st.hr[rows], st.hvpref[rows] = sc[..., S.RAD], sc[..., S.VPREF]
        st.human_times[rows] = 0
        rr = env_for_cfg.spec().robot_row()
        st.rpx[rows], st.rpy[rows], st.rgx[rows], st.rgy[rows] = rr[S.PX], rr[S.PY], rr[S.GX], rr[S.GY]
        st.rvx[rows], st.rvy[rows], st.rr[rows], st.gtime[rows] = 0.0, 0.0, rr[S.RAD], 0.0
    load(np.arange(n), env_ids % 500)
    next_case = (env_ids + 1) % 500
    cfg = H.oracle_cfg_for(env_for_cfg)
    episodes = 0
    for t in range(T):
        a = acts_np[t][env_ids]
        ref = cport.env_step(cfg, st, a[:, 0].copy(), a[:, 1].copy(), update=True)
        d = np.nonzero(ref["done"])[0]
        if len(d):
            load(d, next_case[d])
            next_case[d] = (next_case[d] + 1) % 500
            episodes += len(d)
    return st, ref, episodes


@pytest.mark.parametrize("N,T", [(5, 1000), (10, 500)])
def test_bench_timed_launch_shapes_against_single_steps_and_oracle(N, T):
    """bench.py's timed configuration ITSELF: 4096 envs from bench.build_env (500-case pool, in-kernel restarts, Explorer
    records), the driver's 20-step table-action sequence repeated to ONE 1000-step mcn_env_rollout launch (N = 5:
    env_rollout_quad_kernel, state in registers) / ONE 500-step looped launch (N = 10: env_step_loop_kernel) -- against
    (a) one mcn_env_step launch per step, every byte of state and records, and (b) the C oracle's trajectory of 64
    sampled envs with the pool restarts replayed on the host, bit for bit."""
    torch = _torch()
    import bench
    dev = torch.device("cuda", 0)
    E, K = 4096, 20
    acts = bench.make_actions(5 + K, E, E, 0, dev)[5:].repeat(T // K, 1, 1).contiguous()     # bench.py: acts_rep
    assert acts.shape[0] == T
    a, pool = bench.build_env(E, N, 0, dev)
    b, _ = bench.build_env(E, N, 0, dev)
    a.rollout(acts)                                           # ONE launch of T steps
    for t in range(T):
        b.step(acts[t])
    torch.cuda.synchronize()
    for k in ("hpos", "hvel", "hgoal", "hrad", "hvpref", "rpos", "rvel", "rgoal", "gtime", "step_rec"):
        assert np.array_equal(getattr(a, k).cpu().numpy().view(np.uint8), getattr(b, k).cpu().numpy().view(np.uint8)), k
    for k in ("state", "fin_return", "fin_time", "fin_info"):
        assert np.array_equal(a.rollout_buffers[k].cpu().numpy().view(np.uint8),
                              b.rollout_buffers[k].cpu().numpy().view(np.uint8)), k
    assert int(a.rollout_buffers["fin_count"].sum().item()) > 4 * E
    # (b) the oracle on 64 envs spread over the batch
    ids = np.arange(64) * 64 + (np.arange(64) % 7)
    st, ref, episodes = _oracle_with_pool_restarts(ids, pool, N, acts.cpu().numpy(), T, a)
    assert episodes > 4 * 64
    got = {k: getattr(a, k).cpu().numpy()[ids] for k in ("hpos", "hvel", "hgoal", "hrad", "rpos", "rvel", "gtime")}
    assert np.array_equal(got["hpos"][..., 0], st.hpx) and np.array_equal(got["hpos"][..., 1], st.hpy)
    assert np.array_equal(got["hvel"][..., 0], st.hvx) and np.array_equal(got["hvel"][..., 1], st.hvy)
    assert np.array_equal(got["hgoal"][..., 0], st.hgx) and np.array_equal(got["hrad"], st.hr)
    assert np.array_equal(got["rpos"][:, 0], st.rpx) and np.array_equal(got["rpos"][:, 1], st.rpy)
    assert np.array_equal(got["rvel"][:, 0], st.rvx) and np.array_equal(got["gtime"], st.gtime)
    assert np.array_equal(a.reward.cpu().numpy()[ids], ref["reward"]) and np.array_equal(a.done.cpu().numpy()[ids], ref["done"])
    assert np.array_equal(a.info.cpu().numpy()[ids], ref["info"]) and np.array_equal(a.dmin.cpu().numpy()[ids], ref["dmin"])


def test_rollout_launch_matches_oracle_trajectory():
    """The fused T-step launch against the C oracle stepped T times (no pool: finished envs keep stepping, which the
    oracle does too)."""
    torch = _torch()
    E, N, T = 513, 5, 40
    env = H.make_vec_env(E, N)
    env.reset("test", test_cases=[i % 500 for i in range(E)])
    st = H.download(env)
    cfg = H.oracle_cfg_for(env)
    rng = np.random.RandomState(3)
    sp, aa = rng.uniform(0, 1, (T, E)), rng.uniform(0, 2 * np.pi, (T, E))
    ax, ay = sp * np.cos(aa), sp * np.sin(aa)
    from modelcrowdnav_amd import _hip
    with _hip.tuned(rollout_fused=1):
        env.rollout(torch.from_numpy(np.stack([ax, ay], -1)).to(env.device))
    for t in range(T):
        ref = cport.env_step(cfg, st, ax[t], ay[t], update=True)
    H.assert_state_equal(H.download(env), st, what="after a %d-step launch" % T)
    assert np.array_equal(env.reward.cpu().numpy(), ref["reward"]) and np.array_equal(env.done.cpu().numpy(), ref["done"])
    assert np.array_equal(env.info.cpu().numpy(), ref["info"]) and np.array_equal(env.dmin.cpu().numpy(), ref["dmin"])
    assert np.array_equal(env.hh_count.cpu().numpy(), ref["hh_count"])
    np.testing.assert_array_equal(env.human_act.cpu().numpy(), ref["human_act"])


def test_rollout_rejects_bad_arguments():
    torch = _torch()
    from modelcrowdnav_amd import _hip
    env = H.make_vec_env(8, 5)
    env.reset("test", test_cases=list(range(8)))
    with pytest.raises(ValueError):
        env.rollout(torch.zeros(3, 7, 2, dtype=torch.float64, device=env.device))
    with pytest.raises(_hip.McnError):
        env.rollout(torch.zeros(0, 8, 2, dtype=torch.float64, device=env.device))


def test_unicycle_robot_matches_oracle_and_reference(golden_dir):
    """Robot with (v, r) actions (agent.py:110-135, crowd_sim.py:353-355).  cos/sin differ between numpy, libm
    and the device by an ulp, so float state is held to 1e-12; masks stay exact on these inputs."""
    torch = _torch()
    rng = np.random.RandomState(77)
    E, N = 500, 5
    env = H.make_vec_env(E, N, kinematics="unicycle")
    st = H.random_state(rng, E, N)
    st.rtheta[:] = rng.uniform(-7, 7, E)
    v, r = rng.uniform(0, 1, E), rng.uniform(-np.pi / 4, np.pi / 4, E)
    H.upload(env, st)
    ob, reward, done, info = env.step(torch.from_numpy(np.stack([v, r], -1)).to(env.device))
    ref_st = st.copy()
    ref = cport.env_step(H.oracle_cfg_for(env), ref_st, v, r, update=True)
    assert np.array_equal(done.cpu().numpy(), ref["done"]) and np.array_equal(info.cpu().numpy(), ref["info"])
    np.testing.assert_allclose(reward.cpu().numpy(), ref["reward"], rtol=0, atol=1e-12)
    got = H.download(env)
    for k in ("rpx", "rpy", "rvx", "rvy", "rtheta", "hpx", "hpy", "gtime"):
        np.testing.assert_allclose(getattr(got, k), getattr(ref_st, k), rtol=0, atol=1e-12, err_msg=k)
    # recorded from the real reference (ModelCrowdSim.step with an ActionRot robot)
    g = np.load(os.path.join(golden_dir, "g2_step_unicycle.npz"))
    for N in np.unique(g["N"]):
        sel = np.where((g["N"] == N) & (g["update"] == 1))[0]
        E, N = len(sel), int(N)
        env = H.make_vec_env(E, N, kinematics="unicycle")
        env.count_hh = False
        st = cport.EnvState(E, N)
        h, rb = g["hum_in"][sel][:, :N], g["rob_in"][sel]
        st.hpx[:], st.hpy[:], st.hvx[:], st.hvy[:], st.hr[:] = h[..., 0], h[..., 1], h[..., 2], h[..., 3], h[..., 4]
        st.rpx[:], st.rpy[:], st.rvx[:], st.rvy[:], st.rr[:] = rb[:, 0], rb[:, 1], rb[:, 2], rb[:, 3], rb[:, 4]
        st.rgx[:], st.rgy[:], st.rtheta[:] = rb[:, 5], rb[:, 6], rb[:, 8]
        st.gtime[:] = g["time"][sel]
        H.upload(env, st)
        gv = torch.from_numpy(g["given_v"][sel][:, :N].copy()).to(env.device)
        ob, reward, done, info = env.step(torch.from_numpy(g["act"][sel]).to(env.device), given_v=gv)
        assert np.array_equal(done.cpu().numpy(), g["done"][sel].astype(np.uint8))
        assert np.array_equal(info.cpu().numpy(), g["info"][sel].astype(np.uint8))
        np.testing.assert_allclose(reward.cpu().numpy(), g["reward"][sel], rtol=0, atol=1e-12)
        ro = g["rob_out"][sel]
        np.testing.assert_allclose(env.rpos.cpu().numpy(), ro[:, 0:2], rtol=0, atol=1e-12)
        np.testing.assert_allclose(env.rvel.cpu().numpy(), ro[:, 2:4], rtol=0, atol=1e-12)
        np.testing.assert_allclose(env.rtheta.cpu().numpy(), ro[:, 8], rtol=0, atol=1e-12)


@pytest.mark.parametrize("kernel", ["lane-per-human", "quad"])
def test_degenerate_configurations_match_oracle(kernel, tuning):
    """Edge cases the reference can reach: humans standing on their goal (zero preferred velocity), everybody at
    rest, agents overlapping or exactly coincident (the float32 solver then divides 0/0 -- the NaNs it produces
    must be the same NaNs), robot exactly on its goal, robot starting inside a human, zero robot action
    (degenerate swept segment).  Bit-exact including NaN positions."""
    tuning(quad_max_envs=0 if kernel == "lane-per-human" else 1 << 30)
    torch = _torch()
    rng = np.random.RandomState(2024)
    E, N = 240, 5
    env = H.make_vec_env(E, N)
    st = H.random_state(rng, E, N)
    blk = E // 8
    s = lambda i: slice(i * blk, (i + 1) * blk)
    st.hgx[s(0)], st.hgy[s(0)] = st.hpx[s(0)], st.hpy[s(0)]                  # humans on their goals
    st.hvx[s(1)] = 0; st.hvy[s(1)] = 0; st.rvx[s(1)] = 0; st.rvy[s(1)] = 0   # everybody at rest
    st.hpx[s(2), 1], st.hpy[s(2), 1] = st.hpx[s(2), 0] + 0.1, st.hpy[s(2), 0]   # two humans overlapping
    st.hpx[s(3), 2], st.hpy[s(3), 2] = st.hpx[s(3), 4], st.hpy[s(3), 4]      # two humans coincident ...
    st.hvx[s(3), 2], st.hvy[s(3), 2] = st.hvx[s(3), 4], st.hvy[s(3), 4]      # ... with equal velocities: 0/0
    st.rgx[s(4)], st.rgy[s(4)] = st.rpx[s(4)], st.rpy[s(4)]                  # robot on its goal
    st.rpx[s(5)], st.rpy[s(5)] = st.hpx[s(5), 3], st.hpy[s(5), 3]            # robot inside a human
    st.hpx[s(6)] = rng.uniform(-30, 30, (blk, N)); st.hpy[s(6)] = rng.uniform(-30, 30, (blk, N))   # beyond neighborDist
    ax, ay = rng.uniform(-1, 1, E), rng.uniform(-1, 1, E)
    ax[s(7)] = 0; ay[s(7)] = 0                                               # zero action
    for i in range(blk):                                                     # ... and human at rest: px == ex
        st.hvx[7 * blk + i, 0] = 0; st.hvy[7 * blk + i, 0] = 0
    for update in (True, False):
        got, ref, ref_st = _step_both(env, st, ax, ay, update)
        for key in ref:
            assert np.array_equal(got[key], ref[key], equal_nan=True), (key, update)
        if update:
            dl = H.download(env)
            for f in H.STATE_FIELDS:
                assert np.array_equal(getattr(dl, f), getattr(ref_st, f), equal_nan=True), f
    # (coincident agents make a NaN half-plane; every comparison with it is false, so the LP skips it and the
    # outputs stay finite -- in the oracle and on the device alike)
    assert np.isfinite(ref["human_act"]).all()
    assert (ref["hh_count"][s(2)] >= 1).all()
    assert (ref["info"][s(5)] == cport.INFO_COLLISION).sum() + (ref["info"][s(5)] == cport.INFO_TIMEOUT).sum() == blk


def test_single_env_and_tiny_batches():
    """E = 1, 2, 3: grids smaller than one wavefront."""
    torch = _torch()
    rng = np.random.RandomState(3)
    for E in (1, 2, 3, 13):
        for N in (1, 5):
            env = H.make_vec_env(E, N)
            st = H.random_state(rng, E, N)
            ax, ay = rng.uniform(-1, 1, E), rng.uniform(-1, 1, E)
            got, ref, ref_st = _step_both(env, st, ax, ay, True)
            for key in ref:
                assert np.array_equal(got[key], ref[key]), (E, N, key)
            H.assert_state_equal(H.download(env), ref_st)


def test_config5_shape_32768x10_exact_and_shard_invariant():
    """BASELINE configs[4] shape on one GPU: 32768 envs x 10 humans (ORCA, 9 neighbours each).
    (a) 6 steps bit-exact against the C oracle for the whole batch; (b) stepping the batch as 8 shards of 4096
    (what 8 ranks do) gives bit-identical state to stepping it whole -- no cross-env coupling anywhere."""
    torch = _torch()
    from modelcrowdnav_amd.envs import scenarios as S
    E, N, T = 32768, 10, 6
    env = H.make_vec_env(E, N)
    env.track_human_times = False; env.export_human_actions = False
    pool = S.scenario_pool(env.spec(), "test", range(64), N, "circle_crossing")
    scen = pool[np.arange(E) % 64]
    rng = np.random.RandomState(5)
    scen[:, :, [S.VX, S.VY]] = rng.uniform(-0.5, 0.5, (E, N, 2))          # decorrelate the 512 copies of a case
    env.load_scenarios(scen)
    st = H.download(env)
    cfg = H.oracle_cfg_for(env)
    acts = rng.uniform(-0.7, 0.7, (T, E, 2))
    shards = []
    for r in range(8):
        sub = H.make_vec_env(4096, N)
        sub.track_human_times = False; sub.export_human_actions = False
        sub.load_scenarios(scen[r * 4096:(r + 1) * 4096])
        shards.append(sub)
    for t in range(T):
        ob, reward, done, info = env.step(torch.from_numpy(acts[t]).to(env.device))
        ref = cport.env_step(cfg, st, acts[t, :, 0].copy(), acts[t, :, 1].copy(), update=True)
        assert np.array_equal(done.cpu().numpy(), ref["done"]) and np.array_equal(info.cpu().numpy(), ref["info"]), t
        assert np.array_equal(reward.cpu().numpy(), ref["reward"]), t
        for r, sub in enumerate(shards):
            sub.step(torch.from_numpy(acts[t, r * 4096:(r + 1) * 4096]).to(sub.device))
    H.assert_state_equal(H.download(env), st, fields=[f for f in H.STATE_FIELDS if f != "human_times"])
    whole = env.hpos.cpu().numpy()
    parts = np.concatenate([s_.hpos.cpu().numpy() for s_ in shards], 0)
    assert np.array_equal(whole, parts)
    assert np.array_equal(env.rpos.cpu().numpy(), np.concatenate([s_.rpos.cpu().numpy() for s_ in shards], 0))


def test_bench_kernel_attribution_matches_the_dispatcher(tuning):
    """bench.expected_kernel() decides which profile rows (PMC traffic, SQ counters) a roofline entry is built from; it
    re-states the library's dispatch rules in Python.  Every launch shape bench.py reports is launched here once and the
    family the dispatcher REALLY took (mcn_last_dispatch) must be the one bench names (with the library's own defaults:
    a run of the suite under MCN_* overrides must not change what this test checks)."""
    torch = _torch()
    import bench
    from modelcrowdnav_amd import _hip
    tuning(force_generic=0, quad_max_envs=-1, quad_split=-1, rollout_fused=-1, rollout_split=-1, step_block=-1,
           pair_stream=-1, lp3_defer=-1)
    dev = torch.device("cuda", 0)
    for E, N, spl in [(4096, 5, 1), (4096, 5, 20), (4096, 5, 1000), (4096, 10, 1), (4096, 10, 500), (32768, 10, 1),
                      (1 << 16, 5, 1), (1 << 18, 10, 1)]:
        env, _ = bench.build_env(E, N, 0, dev)
        acts = bench.make_actions(2, E, E, 0, dev)
        if spl > 1:
            env.rollout(acts)
        else:
            env.step(acts[0])
        torch.cuda.synchronize()
        assert _hip.last_dispatch() == bench.expected_kernel(E, N, False, spl), (E, N, spl, _hip.last_dispatch())
        del env
    for E, N in [(4096, 5), (1 << 18, 5), (1 << 17, 10)]:                   # the given-velocity (pairwise) sweep
        env, _ = bench.build_env(E, N, 0, dev)
        env.count_hh = False
        acts = bench.make_actions(1, E, E, 0, dev)
        env.step(acts[0], given_v=torch.zeros(E, N, 2, dtype=torch.float64, device=dev))
        torch.cuda.synchronize()
        assert _hip.last_dispatch() == bench.expected_kernel(E, N, True, 1), (E, N, _hip.last_dispatch())
        del env
