"""GPU: the callers either side of the step kernel -- batched Explorer, the E = 1 gym views driven like the
reference's drivers drive them (test.py:64-109, explorer.py:54-125), ModelCrowdSim with a world model."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import cport  # noqa: E402
from tests import helpers as H  # noqa: E402


def _goal_seeking(env, t):
    """Deterministic, state-only robot policy whose arithmetic is exact on any IEEE machine: move at +-0.6 along
    each axis according to the sign of the remaining goal offset (0 inside a 0.2 band)."""
    import torch
    d = env.rgoal - env.rpos
    v, z = torch.full_like(d, 0.6), torch.zeros_like(d)          # float64 constants (python scalars would be f32)
    return (torch.where(d > 0.2, v, z) - torch.where(d < -0.2, v, z)).contiguous()


def _oracle_episode(scen, gamma=0.9):
    st = cport.EnvState(1, scen.shape[0])
    st.hpx[0], st.hpy[0], st.hgx[0], st.hgy[0] = scen[:, 0], scen[:, 1], scen[:, 2], scen[:, 3]
    st.hr[0], st.hvpref[0] = scen[:, 7], scen[:, 8]
    st.rpy[0], st.rgy[0], st.rr[0] = -4.0, 4.0, 0.3
    cfg = cport.default_cfg()
    rewards, too_close, min_dist = [], 0, 0.0
    while True:
        dx, dy = st.rgx[0] - st.rpx[0], st.rgy[0] - st.rpy[0]
        ax = 0.6 if dx > 0.2 else (-0.6 if dx < -0.2 else 0.0)
        ay = 0.6 if dy > 0.2 else (-0.6 if dy < -0.2 else 0.0)
        out = cport.env_step(cfg, st, np.array([ax]), np.array([ay]))
        rewards.append(float(out["reward"][0]))
        if out["info"][0] == cport.INFO_DANGER:                      # explorer.py:88-90
            too_close += 1
            min_dist += float(out["dmin"][0])
        if out["done"][0]:
            tm = 25.0 if out["info"][0] == cport.INFO_TIMEOUT else float(st.gtime[0])
            ret = sum([pow(gamma, t * 0.25 * 1.0) * r for t, r in enumerate(rewards)])
            return ret, int(out["info"][0]), tm, too_close, min_dist


def test_vec_explorer_equals_sequential_loop():
    """k episodes over E < k envs with in-kernel auto-reset == the reference's one-at-a-time loop."""
    from modelcrowdnav_amd.envs import scenarios as S
    from modelcrowdnav_amd.rollout import VecExplorer
    E, N, k = 32, 5, 70
    env = H.make_vec_env(E, N)
    env.track_human_times = False; env.export_human_actions = False
    ex = VecExplorer(env, env.robot, gamma=0.9, policy=object())
    avg, sr, cr, tr, nav = ex.run_k_episodes(k, "test", action_fn=_goal_seeking, returnNav=True)
    want = [_oracle_episode(S.scenario_for_case(env.spec(), "test", c, N, "circle_crossing")) for c in range(k)]
    got = ex.last_records
    assert got["infos"] == [w[1] for w in want]
    bad = [i for i in range(k) if got["returns"][i] != want[i][0] or got["times"][i] != want[i][2]]
    assert not bad, (bad, [got["returns"][i] for i in bad], [want[i][0] for i in bad])
    assert avg == sum([w[0] for w in want]) / k
    assert len(set(got["infos"])) > 1, "test should exercise more than one outcome"
    assert env.case_counter["test"] == k % 500
    # the "too close" statistics cover exactly the k episodes (k = 70 over 32 envs: 6 envs play three, 26 play two,
    # and every env keeps stepping until the slowest one is done)
    assert got["danger_steps"] == sum(w[3] for w in want) > 0
    assert abs(got["danger_dist_sum"] - sum(w[4] for w in want)) < 1e-9


def test_vec_explorer_case_counter_wraps_mid_run():
    """The reference's counter wraps at case_size (crowd_sim.py:296): 20 validation episodes starting at case 90 of 100
    play cases 90..99, 0..9 -- over 8 envs that is an uneven walk through the case list (3 rounds)."""
    from modelcrowdnav_amd.envs import scenarios as S
    from modelcrowdnav_amd.rollout import VecExplorer
    E, N, k = 8, 5, 20
    env = H.make_vec_env(E, N)
    env.track_human_times = False; env.export_human_actions = False
    assert env.case_size["val"] == 100
    env.case_counter["val"] = 90
    ex = VecExplorer(env, env.robot, gamma=0.9, policy=object())
    avg, sr, cr, tr = ex.run_k_episodes(k, "val", action_fn=_goal_seeking)
    cases = [(90 + i) % 100 for i in range(k)]
    want = [_oracle_episode(S.scenario_for_case(env.spec(), "val", c, N, "circle_crossing")) for c in cases]
    got = ex.last_records
    assert got["infos"] == [w[1] for w in want]
    assert got["returns"] == [w[0] for w in want] and got["times"] == [w[2] for w in want]
    assert avg == sum(w[0] for w in want) / k
    assert env.case_counter["val"] == 10
    assert got["danger_steps"] == sum(w[3] for w in want)


@pytest.mark.parametrize("N", [5, 10])
def test_vec_explorer_action_sequence_uses_fused_rollout(N):
    """A pre-drawn action sequence run through mcn_env_rollout chunks gives the records of the per-step loop: 5 humans
    through the quad-parallel rollout kernel, 10 through the looped step kernel (env_step_loop_kernel)."""
    import torch
    from modelcrowdnav_amd.rollout import VecExplorer
    E, k, T = 48, 100, 330
    rng = np.random.RandomState(5)
    sp, aa = rng.uniform(0, 1, (T, E)), rng.uniform(0, 2 * np.pi, (T, E))
    recs = []
    for fused in (True, False):
        env = H.make_vec_env(E, N)
        env.track_human_times = False; env.export_human_actions = False
        seq = torch.from_numpy(np.stack([sp * np.cos(aa), sp * np.sin(aa)], -1)).to(env.device)
        ex = VecExplorer(env, env.robot, gamma=0.9, policy=object())
        if fused:
            out = ex.run_k_episodes(k, "val", action_seq=seq)
        else:
            out = ex.run_k_episodes(k, "val", action_fn=lambda env_, t: seq[t])
        recs.append((out, ex.last_records))
    assert recs[0][0] == recs[1][0]
    for key in ("returns", "infos", "times", "danger_steps"):
        assert recs[0][1][key] == recs[1][1][key], key
    assert abs(recs[0][1]["danger_dist_sum"] - recs[1][1]["danger_dist_sum"]) < 1e-9 and recs[0][1]["danger_steps"] > 0
    assert len(set(recs[0][1]["infos"])) > 1


def test_crowdsim_e1_with_orca_robot_matches_oracle():
    """BASELINE config 1 plumbing: CrowdSim gym surface, ORCA humans, ORCA robot (test.py --policy orca)."""
    import torch
    from modelcrowdnav_amd import configs
    from modelcrowdnav_amd.envs import CrowdSim
    from modelcrowdnav_amd.envs.utils.robot import Robot
    from modelcrowdnav_amd.envs.policy.policy_factory import policy_factory
    from modelcrowdnav_amd.envs import scenarios as S
    cfg = configs.env_config()
    env = CrowdSim()
    env.configure(cfg)
    robot = Robot(cfg, "robot")
    pol = policy_factory["orca"]()
    pol.configure(cfg)
    robot.set_policy(pol)
    env.set_robot(robot)
    pol.set_phase("test"); pol.set_device(torch.device("cuda", 0)); pol.set_env(env)
    for case in (0, 3):
        ob = env.reset("test", case)
        scen = S.scenario_for_case(S.ScenarioSpec(), "test", case, 5, "circle_crossing")
        st = cport.EnvState(1, 5)
        st.hpx[0], st.hpy[0], st.hgx[0], st.hgy[0] = scen[:, 0], scen[:, 1], scen[:, 2], scen[:, 3]
        st.hr[0], st.hvpref[0] = scen[:, 7], scen[:, 8]
        st.rpy[0], st.rgy[0], st.rr[0] = -4.0, 4.0, 0.3
        ocfg = cport.default_cfg()
        done, steps = False, 0
        while not done:
            action = robot.act(ob)
            # oracle robot: ORCA over all humans (orca.py:82-132), self radius r + 0.01, max speed v_pref
            opos = np.stack([st.hpx[0], st.hpy[0]], 1); ovel = np.stack([st.hvx[0], st.hvy[0]], 1)
            ov = cport.orca_agent((st.rpx[0], st.rpy[0]), (st.rvx[0], st.rvy[0]), np.float32(0.3 + 0.01), 1.0,
                                  (np.float32(st.rgx[0] - st.rpx[0]), np.float32(st.rgy[0] - st.rpy[0])),
                                  opos, ovel, (st.hr[0] + 0.01).astype(np.float32))
            assert (np.float32(action.vx), np.float32(action.vy)) == ov, (case, steps)
            ob, reward, done, info = env.step(action)
            ref = cport.env_step(ocfg, st, np.array([float(ov[0])]), np.array([float(ov[1])]))
            assert reward == ref["reward"][0] and done == bool(ref["done"][0]) and info.code == ref["info"][0]
            assert [o.px for o in ob] == st.hpx[0].tolist() and [o.vy for o in ob] == st.hvy[0].tolist()
            assert env.global_time == st.gtime[0]
            steps += 1
        assert steps > 10
        assert len(env.states) == steps


def test_get_human_times_matches_oracle_simulation():
    """crowd_sim.py:219-258: after the robot has arrived, one centralised all-ORCA simulation to the end (robot = agent
    0, humans after it, everybody at their own radius / v_pref, pref velocity normalised beyond 1 m, float32 positions
    advanced inside the simulator) -- one mcn_orca_batch launch per step -- against the same loop driven by the C
    oracle's solver.  First-arrival times, final positions and the appended states must agree exactly."""
    import torch
    from modelcrowdnav_amd import configs
    from modelcrowdnav_amd.envs import CrowdSim
    from modelcrowdnav_amd.envs.utils.robot import Robot
    from modelcrowdnav_amd.envs.policy.policy_factory import policy_factory
    cfg = configs.env_config()
    env = CrowdSim()
    env.configure(cfg)
    robot = Robot(cfg, "robot")
    pol = policy_factory["orca"]()
    pol.configure(cfg)
    robot.set_policy(pol)
    env.set_robot(robot)
    pol.set_phase("test"); pol.set_device(torch.device("cuda", 0)); pol.set_env(env)
    env.reset("test", 7)
    with pytest.raises(ValueError):
        env.get_human_times()                       # 'Episode is not done yet'
    for _ in range(6):                              # a few real steps so that velocities are not all zero
        env.step(robot.act([h.get_observable_state() for h in env.humans]))
    robot.set_position(robot.get_goal_position())   # the robot has arrived
    agents = [robot] + env.humans
    f32 = np.float32
    pos = np.array([a.get_position() for a in agents]).astype(f32)
    vel = np.array([a.get_velocity() for a in agents]).astype(f32)
    pos64 = np.array([a.get_position() for a in agents], np.float64)       # what the agents' own fields hold
    goal = np.array([a.get_goal_position() for a in agents], np.float64)
    rad = np.array([a.radius for a in agents]).astype(f32)
    vmax = np.array([a.v_pref for a in agents]).astype(f32)
    times, t, n_states = list(env.human_times), env.global_time, len(env.states)
    want_states = 0
    while not all(times):
        pref = goal - pos64
        for i in range(len(agents)):
            nrm = np.linalg.norm(pref[i])
            if nrm > 1:
                pref[i] /= np.linalg.norm(pref[i])
        new = np.zeros_like(vel)
        for i in range(len(agents)):
            oth = [j for j in range(len(agents)) if j != i]
            new[i] = cport.orca_agent(pos[i], vel[i], rad[i], vmax[i], pref[i].astype(f32), pos[oth], vel[oth], rad[oth],
                                      neighbor_dist=10.0, max_neighbors=10, time_horizon=5.0, time_step=0.25)
        vel = new.astype(f32)
        pos = (pos + vel * f32(0.25)).astype(f32)
        t += 0.25
        for i in range(1, len(agents)):
            if times[i - 1] == 0 and np.linalg.norm(pos64[i] - goal[i]) < agents[i].radius:
                times[i - 1] = t
        pos64 = pos.astype(np.float64)
        want_states += 1
        assert want_states < 4000
    got = env.get_human_times()
    assert got == times and all(x > 0 for x in got)
    assert env.global_time == t and len(env.states) == n_states + want_states
    assert [h.get_position() for h in env.humans] == [tuple(p) for p in pos64[1:].tolist()]


@pytest.mark.parametrize("visible", [False, True])
def test_crowdsim_e1_orca_robot_and_human_times_match_reference(visible, golden_dir):
    """g16_orca_robot.npz = BASELINE config 1 as the REAL reference runs it (test.py:64-109 with --policy orca; ORCA
    through the rvo2 stand-in): this build's CrowdSim + ORCA robot takes the same float32 actions, collects the same
    rewards / outcomes and agent states step by step, and get_human_times() (crowd_sim.py:219-258, one mcn_orca_batch
    launch per simulated step) returns the same first-arrival times, end positions, clock and number of states."""
    import torch
    from modelcrowdnav_amd import configs
    from modelcrowdnav_amd.envs import CrowdSim
    from modelcrowdnav_amd.envs.utils.robot import Robot
    from modelcrowdnav_amd.envs.policy.policy_factory import policy_factory
    g = np.load(os.path.join(golden_dir, "g16_orca_robot.npz"))
    cfg = configs.env_config(**{"robot.visible": "true" if visible else "false"})
    env = CrowdSim()
    env.configure(cfg)
    robot = Robot(cfg, "robot")
    pol = policy_factory["orca"]()
    pol.configure(cfg)
    robot.set_policy(pol)
    env.set_robot(robot)
    pol.set_phase("test"); pol.set_device(torch.device("cuda", 0)); pol.set_env(env)

    def rows():
        r = env.robot
        rob = [r.px, r.py, r.vx, r.vy, r.radius, r.gx, r.gy, r.v_pref, r.theta]
        hum = [[h.px, h.py, h.vx, h.vy, h.radius, h.gx, h.gy, h.v_pref, h.theta] for h in env.humans]
        return np.array(rob), np.array(hum)
    arrived = 0
    for case in (0, 3, 6, 11):
        key = "v%d_c%d_" % (visible, case)
        ob = env.reset("test", case)
        A, I = g[key + "actions"], g[key + "info"]
        for t in range(A.shape[0]):
            action = robot.act(ob)
            assert (action.vx, action.vy) == (A[t, 0], A[t, 1]), (case, t)
            ob, reward, done, info = env.step(action)
            assert reward == g[key + "rewards"][t] and info.code == I[t] and done == (t == A.shape[0] - 1), (case, t)
            rob, hum = rows()
            assert np.array_equal(np.concatenate([rob, hum.ravel()]), g[key + "states"][t]), (case, t)
        assert env.global_time == float(g[key + "time"])
        assert np.array_equal(np.array(env.human_times, np.float64), g[key + "human_times_step"])
        if key + "human_times" in g.files:
            arrived += 1
            assert np.array_equal(np.array(env.get_human_times(), np.float64), g[key + "human_times"]), case
            rob, hum = rows()
            assert np.array_equal(rob, g[key + "end_rob"]) and np.array_equal(hum, g[key + "end_hum"]), case
            assert env.global_time == float(g[key + "end_time"]) and len(env.states) == int(g[key + "n_states"])
        else:
            with pytest.raises(ValueError):
                env.get_human_times()               # 'Episode is not done yet' (the robot collided on the way)
    assert arrived >= 2


def test_crowdsim_e1_sarl_episode_matches_reference(golden_dir):
    """G7: reference CrowdSim + SARL (seeded weights) episodes; this build's CrowdSim + SARL must take the same
    actions and collect the same rewards step by step."""
    import torch
    from modelcrowdnav_amd import configs
    from modelcrowdnav_amd.envs import CrowdSim
    from modelcrowdnav_amd.envs.utils.robot import Robot
    from modelcrowdnav_amd.policy.policy_factory import policy_factory
    g = np.load(os.path.join(golden_dir, "g7_episode.npz"))
    for seed, N in ((0, 5), (0, 10)):
        cfg = configs.env_config(**{"sim.human_num": N})
        env = CrowdSim()
        env.configure(cfg)
        robot = Robot(cfg, "robot")
        pol = policy_factory["sarl"]()
        pol.configure(configs.policy_config())
        pol.kinematics = "holonomic"
        pref = "ep%d_N%d_w__" % (seed, N)
        pol.model.load_state_dict({k[len(pref):].replace("__", "."): torch.from_numpy(g[k]) for k in g.files
                                   if k.startswith(pref)})
        robot.set_policy(pol)
        env.set_robot(robot)
        pol.set_phase("test"); pol.set_device(torch.device("cuda", 0)); pol.set_env(env)
        for case in (0, 1):
            key = "ep%d_N%d_c%d_" % (seed, N, case)
            ob = env.reset("test", case)
            done, t, rewards = False, 0, []
            while not done:
                action = robot.act(ob)
                want = g[key + "actions"][t]
                assert (action.vx, action.vy) == tuple(want), (seed, N, case, t)
                ob, reward, done, info = env.step(action)
                assert reward == g[key + "rewards"][t] and info.code == g[key + "info"][t]
                rewards.append(reward)
                t += 1
            assert t == len(g[key + "rewards"])
            ret = sum([pow(0.9, k * robot.time_step * robot.v_pref) * r for k, r in enumerate(rewards)])
            assert ret == float(g[key + "return"])
            assert env.global_time == float(g[key + "time"])


@pytest.mark.parametrize("visible", [False, True])
def test_query_env_lookahead_matches_reference(visible, golden_dir):
    """G11: `query_env = true` (multi_human_rl.py:37-38, cadrl.py:158-159).  The reference evaluates each of the 81
    candidate actions through env.onestep_lookahead; here one mcn_env_step(update = 0) supplies the humans' reaction, one
    given-velocity mcn_env_step over the 81 (env, action) pairs the rewards, and mcn_sarl_lookahead_env the values.
    Per step: the 81 action values to 1e-5, the chosen action, reward and info -- with an invisible and a visible robot;
    plus the batched form over several envs against the E = 1 path."""
    import torch
    from modelcrowdnav_amd import configs
    from modelcrowdnav_amd.envs import CrowdSim
    from modelcrowdnav_amd.envs.utils.robot import Robot
    from modelcrowdnav_amd.policy.policy_factory import policy_factory
    g = np.load(os.path.join(golden_dir, "g11_queryenv.npz"))
    cfg = configs.env_config(**{"robot.visible": "true" if visible else "false"})
    env = CrowdSim()
    env.configure(cfg)
    robot = Robot(cfg, "robot")
    pol = policy_factory["sarl"]()
    pol.configure(configs.policy_config())
    pol.kinematics = "holonomic"
    pol.model.load_state_dict({k[3:].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith("w__")})
    robot.set_policy(pol)
    env.set_robot(robot)
    pol.set_phase("test"); pol.set_device(torch.device("cuda", 0)); pol.set_env(env)
    pol.query_env = True
    for case in (0, 5):
        key = "vis%d_c%d_" % (int(visible), case)
        ob = env.reset("test", case)
        for t in range(len(g[key + "rewards"])):
            action = robot.act(ob)
            want_v = g[key + "values"][t]
            np.testing.assert_allclose(np.array(pol.action_values), want_v, rtol=0, atol=1e-5, err_msg="%s step %d" % (key, t))
            top2 = np.sort(want_v)[-2:]
            if top2[1] - top2[0] > 2e-5:
                assert (action.vx, action.vy) == tuple(g[key + "actions"][t]), (key, t)
            # follow the reference's trajectory even where a near-tie was broken differently
            from modelcrowdnav_amd.envs.utils.action import ActionXY
            ob, reward, done, info = env.step(ActionXY(*g[key + "actions"][t]))
            assert reward == g[key + "rewards"][t] and info.code == g[key + "info"][t], (key, t)
    # batched: predict_batch(query_env) over a VecCrowdSim in the states of several test cases == the E = 1 path
    from tests import helpers as H
    E = 6
    venv = H.make_vec_env(E, 5, robot_visible=visible)
    venv.reset("test", test_cases=[3, 4, 5, 6, 7, 8])
    venv.robot.policy = pol                     # (only v_pref / visibility are read from the robot)
    rng = np.random.RandomState(4)
    for _ in range(5):                           # move off the initial layout
        venv.step(torch.from_numpy(rng.uniform(-0.5, 0.5, (E, 2))).to(venv.device))
    _, best, values = pol.predict_batch(venv, want_values=True)
    best, values = best.clone(), values.clone()                     # (the policy reuses its output buffers)
    pol.query_env = False
    _, _, values_cv = pol.predict_batch(venv, want_values=True)
    values_cv = values_cv.clone()
    pol.query_env = True
    assert float((values - values_cv).abs().max()) > 1e-6          # the two look-aheads really differ
    for e in range(E):
        single = H.make_vec_env(1, 5, robot_visible=visible)
        single.load_scenarios(np.zeros((1, 5, 9)))
        for name in ("hpos", "hvel", "hgoal", "hrad", "hvpref", "rpos", "rvel", "rgoal", "rrad", "rvpref", "rtheta", "gtime"):
            getattr(single, name).copy_(getattr(venv, name)[e:e + 1])
        _, b1, v1 = pol.predict_batch(single, want_values=True)
        assert torch.equal(v1[0], values[e]) and int(b1[0]) == int(best[e])


@pytest.mark.parametrize("world_on", ["cuda", "cpu"])
def test_look_ahead_in_sim_matches_reference(world_on, golden_dir):
    """g18_lookahead_in_sim.npz = the REAL reference's CrowdSim with `look_ahead_in_sim = true`: env.onestep_lookahead
    goes to step_in_sim (crowd_sim.py:325-329,633-696), whose humans are moved by a bare MlpWorld module, and SARL with
    `query_env` evaluates its 81 actions through it.  Per step: the look-ahead's observation / reward / info for one
    fixed action, the 81 action values (1e-5), the chosen action; the real steps (ORCA humans) stay exact.  The module
    may sit on the GPU or -- as the reference's drivers leave it -- on the host."""
    import torch
    from modelcrowdnav_amd import configs
    from modelcrowdnav_amd.envs import CrowdSim
    from modelcrowdnav_amd.envs.utils.action import ActionXY
    from modelcrowdnav_amd.envs.utils.robot import Robot
    from modelcrowdnav_amd.policy.policy_factory import policy_factory
    from modelcrowdnav_amd.policy.world_model import MlpWorld
    g = np.load(os.path.join(golden_dir, "g18_lookahead_in_sim.npz"))
    load = lambda pref: {k[len(pref):].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith(pref)}
    cfg = configs.env_config()
    env = CrowdSim()
    env.configure(cfg)
    robot = Robot(cfg, "robot")
    pol = policy_factory["sarl"]()
    pol.configure(configs.policy_config())
    pol.kinematics = "holonomic"
    pol.model.load_state_dict(load("w__"))
    robot.set_policy(pol)
    env.set_robot(robot)
    pol.set_phase("test"); pol.set_device(torch.device("cuda", 0)); pol.set_env(env)
    pol.query_env = True
    world = MlpWorld(5)
    world.load_state_dict(load("world__"))
    world.eval().to(world_on)
    env.reset("test", 0)
    with pytest.raises(AttributeError):
        env.look_ahead_in_sim = True
        env.onestep_lookahead(ActionXY(0.0, 0.0))                     # no sim_world yet
    env.sim_world = world
    for case in (1, 4):
        key = "c%d_" % case
        ob = env.reset("test", case)
        for t in range(len(g[key + "rewards"])):
            action = robot.act(ob)
            want_v = g[key + "values"][t]
            np.testing.assert_allclose(np.array(pol.action_values), want_v, rtol=0, atol=1e-5, err_msg="%s step %d" % (key, t))
            top2 = np.sort(want_v)[-2:]
            if top2[1] - top2[0] > 2e-5:
                assert (action.vx, action.vy) == tuple(g[key + "actions"][t]), (key, t)
            before = [(h.px, h.py, h.vx, h.vy) for h in env.humans] + [env.global_time]
            lob, lr, ld, linfo = env.onestep_lookahead(ActionXY(0.3, -0.2))
            got = np.array([[o.px, o.py, o.vx, o.vy, o.radius] for o in lob])
            np.testing.assert_allclose(got, g[key + "look_obs"][t], rtol=0, atol=2e-6)      # float32 module outputs
            assert abs(lr - g[key + "look_reward"][t]) < 1e-12 and linfo.code == g[key + "look_info"][t], (key, t)
            assert before == [(h.px, h.py, h.vx, h.vy) for h in env.humans] + [env.global_time]    # nothing mutated
            ob, reward, done, info = env.step(ActionXY(*g[key + "actions"][t]))
            assert reward == g[key + "rewards"][t] and info.code == g[key + "info"][t], (key, t)


def test_model_crowd_sim_with_sgan_world(golden_dir):
    """BASELINE config 4 shape (reduced E): VecModelCrowdSim + VecSGANWorld + SARL robot; the env must move the
    humans exactly by the world model's velocities, and the E = 1 ModelCrowdSim view must agree with env 0."""
    import torch
    from modelcrowdnav_amd import configs
    from modelcrowdnav_amd.envs import VecModelCrowdSim
    from modelcrowdnav_amd.policy.world_model import VecSGANWorld, generator_from_arrays
    from modelcrowdnav_amd.policy.sarl import SARL
    dev = torch.device("cuda", 0)
    E, N = 64, 10
    env = H.make_vec_env(E, N, cls=VecModelCrowdSim)
    env.export_human_actions = True
    env.reset("test", test_cases=list(range(E)))
    gen = generator_from_arrays(np.load(os.path.join(golden_dir, "g6_sgan.npz")), "p", dev)
    world = VecSGANWorld(gen, E, N, dev, time_step=env.time_step, seed=0)
    world.init_constant_velocity(env.hpos, env.hvel)
    env.sim_world = world
    torch.manual_seed(0)
    pol = SARL(); pol.configure(configs.policy_config()); pol.kinematics = "holonomic"
    pol.set_device(dev); pol.set_phase("test"); pol.time_step = env.time_step
    for t in range(12):
        before = env.hpos.clone()
        actions, _ = pol.predict_batch(env)
        ob, reward, done, info = env.step(actions)
        vel = world.out_vel
        assert torch.equal(env.hvel, vel)
        assert torch.equal(env.hpos, before + vel * env.time_step)
        assert torch.equal(env.hh_count, torch.zeros_like(env.hh_count))         # ModelCrowdSim does not count
    assert float(env.gtime[0]) == 12 * 0.25
    # prefetch_world(): the world model on a side stream, overlapping the look-ahead -- the same trajectory, bit for bit
    def run(prefetch):
        e2 = H.make_vec_env(E, N, cls=VecModelCrowdSim)
        e2.reset("test", test_cases=list(range(E)))
        w2 = VecSGANWorld(gen, E, N, dev, time_step=e2.time_step, seed=0)
        w2.fixed_noise = torch.randn(E, 8, generator=torch.Generator().manual_seed(3)).to(dev)
        w2.init_constant_velocity(e2.hpos, e2.hvel)
        e2.sim_world = w2
        for t in range(10):
            if prefetch:
                e2.prefetch_world()
            a, _ = pol.predict_batch(e2)
            e2.step(a)
        torch.cuda.synchronize()
        return e2.hpos.clone(), e2.rpos.clone(), e2.step_rec.clone()
    for x, y in zip(run(False), run(True)):
        assert torch.equal(x, y)


def _reference_targets(states, rewards, dones, infos, gamma_bar, il, target_model=None, rounds=None):
    """explorer.py:153-186 written out naively per env and episode (float64 on the host)."""
    import torch
    T, E = rewards.shape
    out_s, out_v = [], []
    for e in range(E):
        start, ep = 0, 0
        for t in range(T):
            if not dones[t, e]:
                continue
            if rounds is not None and ep >= rounds:
                break
            seg = list(range(start, t + 1))
            if infos[t, e] in (2, 3):                       # ReachGoal / Collision only
                for i in seg:
                    if il:
                        v = sum([pow(gamma_bar, max(k - i, 0)) * rewards[k, e] * (1 if k >= i else 0) for k in seg])
                    elif i == seg[-1]:
                        v = rewards[i, e]
                    else:
                        with torch.no_grad():
                            nv = float(target_model(states[i + 1, e].unsqueeze(0)).item())
                        v = rewards[i, e] + gamma_bar * nv
                    out_s.append(states[i, e]); out_v.append(v)
            start, ep = t + 1, ep + 1
    return out_s, np.array(out_v)


@pytest.mark.parametrize("il", [True, False])
def test_update_memory_value_targets(il):
    """Batched update_memory (imitation learning with the ORCA robot / RL with SARL and a target network)
    against the reference's per-episode formulae evaluated on the recorded traces."""
    import torch
    from modelcrowdnav_amd import configs
    from modelcrowdnav_amd.envs.policy.policy_factory import policy_factory
    from modelcrowdnav_amd.policy.sarl import SARL
    from modelcrowdnav_amd.rollout import VecExplorer
    from modelcrowdnav_amd.utils.memory import ReplayMemory
    dev = torch.device("cuda", 0)
    E, N, k = 16, 5, 16
    env = H.make_vec_env(E, N)
    env.track_human_times = False; env.export_human_actions = False
    torch.manual_seed(0)
    sarl = SARL(); sarl.configure(configs.policy_config()); sarl.kinematics = "holonomic"
    sarl.set_device(dev); sarl.set_phase("train"); sarl.time_step = env.time_step
    mem = ReplayMemory(20000, device=dev)
    if il:
        orca = policy_factory["orca"]()
        orca.multiagent_training = True
        orca.safety_space = 0.15                 # train.config imitation_learning.safety_space (invisible robot)
        env.robot.set_policy(orca)
        ex = VecExplorer(env, env.robot, gamma=0.9, policy=orca, memory=mem, target_policy=sarl)
    else:
        sarl.multiagent_training = True
        env.robot.set_policy(sarl)
        ex = VecExplorer(env, env.robot, gamma=0.9, policy=sarl, memory=mem)
        ex.update_target_model(sarl.model)
    # record the same traces through a thin spy on the explorer's own data path
    rec = {}
    orig = ex._value_targets

    def spy(states, rewards, dones, infos, imitation_learning):
        rec.update(states=states.clone(), rewards=rewards.clone(), dones=dones.clone(), infos=infos.clone())
        return orig(states, rewards, dones, infos, imitation_learning)
    ex._value_targets = spy
    # an untrained SARL never finishes an episode (every case times out and timeouts do not feed the memory), so
    # the RL case drives the robot with the exact goal-seeking rule; states / targets still come from SARL
    ex.run_k_episodes(k, "train", update_memory=True, imitation_learning=il,
                      action_fn=None if il else _goal_seeking)
    assert len(mem) > 0
    gbar = pow(0.9, 0.25 * 1.0)
    want_s, want_v = _reference_targets(rec["states"].cpu(), rec["rewards"].cpu().numpy(), rec["dones"].cpu().numpy(),
                                        rec["infos"].cpu().numpy(), gbar, il,
                                        target_model=None if il else ex.target_model.cpu())
    assert len(want_v) == len(mem)
    got_v = torch.stack([mem[i][1] for i in range(len(mem))]).cpu().numpy()[:, 0]
    got_s = torch.stack([mem[i][0] for i in range(len(mem))]).cpu()
    np.testing.assert_allclose(got_v, want_v.astype(np.float32), rtol=0, atol=2e-6)
    assert torch.equal(got_s, torch.stack(want_s))
    if il:
        assert (rec["infos"].cpu().numpy()[rec["dones"].cpu().numpy()] == 2).sum() > 0, "ORCA robot should reach goals"


def test_reference_driver_sequence_through_dropin():
    """The call sequence of crowd_nav/test.py:52-109 and train.py:120-160 with the reference's own import paths
    (after dropin.install()): gym.make, configure, Robot, policy_factory, Explorer.run_k_episodes, imitation-
    learning memory fill.  The sequential Explorer (E = 1 env) and the batched VecExplorer must report the same
    episodes."""
    import torch
    import modelcrowdnav_amd.dropin as dropin
    dropin.install()
    import gym
    from crowd_nav.policy.policy_factory import policy_factory
    from crowd_nav.utils.explorer import Explorer
    from crowd_nav.utils.memory import ReplayMemory
    from crowd_sim.envs.utils.robot import Robot
    from crowd_sim.envs.policy.orca import ORCA
    from modelcrowdnav_amd import configs
    from modelcrowdnav_amd.rollout import VecExplorer
    device = torch.device("cuda:0")
    env_config = configs.env_config()
    policy = policy_factory["orca"]()
    policy.configure(env_config)
    env = gym.make("CrowdSim-v0")
    env.configure(env_config)
    robot = Robot(env_config, "robot")
    robot.set_policy(policy)
    env.set_robot(robot)
    sarl = policy_factory["sarl"]()
    sarl.configure(configs.policy_config()); sarl.kinematics = "holonomic"; sarl.set_device(device)
    memory = ReplayMemory(5000)
    explorer = Explorer(env, robot, device, memory, gamma=0.9, target_policy=sarl)
    policy.set_phase("test"); policy.set_device(device); policy.set_env(env)
    assert isinstance(robot.policy, ORCA)
    robot.policy.safety_space = 0
    k = 4
    explorer.batched = False                  # the sequential E = 1 loop (what k = 1 calls and epsilon-greedy training get)
    avg, sr, cr, tr = explorer.run_k_episodes(k, "test", update_memory=True, imitation_learning=True, print_failure=True)
    assert not explorer.last_run_batched
    assert abs(sr + cr + tr - 1) < 1e-12 and len(memory) > 0
    assert env.case_counter["test"] == k
    state, value = memory[0]
    assert tuple(state.shape) == (5, 13) and tuple(value.shape) == (1,)
    # the same call as the drivers make it (k > 1 goes to the batched explorer behind the same class): same statistics,
    # same memory rows in the same order
    explorer.batched = True
    explorer.memory = ReplayMemory(5000)
    env.case_counter["test"] = 0
    out_b = explorer.run_k_episodes(k, "test", update_memory=True, imitation_learning=True, print_failure=True)
    assert explorer.last_run_batched and env.case_counter["test"] == k
    assert tuple(out_b[1:]) == (sr, cr, tr) and abs(out_b[0] - avg) < 1e-12
    assert len(explorer.memory) == len(memory)
    for i in range(len(memory)):
        np.testing.assert_allclose(explorer.memory[i][0].cpu().numpy(), memory[i][0].cpu().numpy(), rtol=2e-6, atol=2e-6)
        assert abs(float(explorer.memory[i][1]) - float(memory[i][1])) < 2e-6
    # the same four cases, batched
    venv = H.make_vec_env(k, 5)
    venv.track_human_times = False; venv.export_human_actions = False
    orca = policy_factory["orca"](); orca.multiagent_training = True; orca.safety_space = 0
    venv.robot.set_policy(orca)
    vmem = ReplayMemory(5000, device=device)
    vex = VecExplorer(venv, venv.robot, gamma=0.9, policy=orca, memory=vmem, target_policy=sarl)
    vavg, vsr, vcr, vtr = vex.run_k_episodes(k, "test", update_memory=True, imitation_learning=True)
    assert (vsr, vcr, vtr) == (sr, cr, tr)
    assert abs(vavg - avg) < 1e-12
    assert len(vmem) == len(memory)
    got = sorted(float(vmem[i][1]) for i in range(len(vmem)))
    want = sorted(float(memory[i][1]) for i in range(len(memory)))
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6)


def test_device_scenario_pool_follows_the_placement_rules():
    """mcn_scenario_pool: the reference's placement rules (crowd_sim.py:165-215) hold for every generated case,
    cases depend on their id only, and the angle / side statistics are those of the host generator."""
    import torch
    E, N, P = 8, 5, 4096
    for rule, randomize in (("circle_crossing", False), ("square_crossing", False), ("circle_crossing", True)):
        env = H.make_vec_env(E, N, **({"env.randomize_attributes": "true"} if randomize else {}))
        pool = env.device_pool(seed=7, first_case=100, count=P, human_num=N, rule=rule)
        again = env.device_pool(seed=7, first_case=100 + 1000, count=64, human_num=N, rule=rule)
        torch.cuda.synchronize()
        pos, goal, rad, vp = (pool[k].cpu().numpy() for k in ("hpos", "hgoal", "hrad", "hvpref"))
        assert np.array_equal(again["hpos"].cpu().numpy(), pos[1000:1064])           # a case is a function of its id
        other = env.device_pool(seed=8, first_case=100, count=64, human_num=N, rule=rule)["hpos"].cpu().numpy()
        assert not np.array_equal(other, pos[:64])
        if randomize:
            assert rad.min() >= 0.3 and rad.max() < 0.5 and vp.min() >= 0.5 and vp.max() < 1.5
            assert abs(rad.mean() - 0.4) < 0.005 and abs(vp.mean() - 1.0) < 0.02
        else:
            assert (rad == 0.3).all() and (vp == 1.0).all()
        gap = rad[:, :, None] + rad[:, None, :] + 0.2
        d = np.linalg.norm(pos[:, :, None] - pos[:, None, :], axis=-1)
        iu = np.triu_indices(N, 1)
        assert (d[:, iu[0], iu[1]] >= gap[:, iu[0], iu[1]]).all()                         # starts clear of each other
        dg = np.linalg.norm(goal[:, :, None] - goal[:, None, :], axis=-1)
        assert (dg[:, iu[0], iu[1]] >= gap[:, iu[0], iu[1]]).all()                        # goals clear of each other
        rob_gap = rad + 0.3 + 0.2
        assert (np.linalg.norm(pos - np.array([0.0, -4.0]), axis=-1) >= rob_gap).all()    # clear of the robot's start
        if rule == "circle_crossing":
            assert np.array_equal(goal, -pos)                                             # antipodal goals
            r = np.linalg.norm(pos, axis=-1)
            assert (np.abs(r - 4.0) <= 0.5 * np.sqrt(2) * vp + 1e-9).all()                 # on the circle +- noise
            ang = np.arctan2(pos[..., 1], pos[..., 0])
            hist, _ = np.histogram(ang, bins=8, range=(-np.pi, np.pi))
            assert hist.min() > 0.6 * hist.mean()                                         # all directions populated
            assert (np.linalg.norm(pos - np.array([0.0, 4.0]), axis=-1) >= rob_gap).all() # and of its goal
        else:
            assert (np.abs(pos[..., 0]) <= 5.0).all() and (np.abs(pos[..., 1]) <= 5.0).all()
            assert (np.sign(pos[..., 0]) == -np.sign(goal[..., 0])).all()                 # cross to the other side
            assert 0.4 < (pos[..., 0] > 0).mean() < 0.6


def test_vec_explorer_with_device_scenarios():
    """k training episodes from device-generated cases: the rollout machinery (in-kernel restart from the device
    pool, per-episode records) runs end to end and is reproducible."""
    from modelcrowdnav_amd.rollout import VecExplorer
    E, N, k = 64, 5, 200
    outs = []
    for _ in range(2):
        env = H.make_vec_env(E, N)
        env.track_human_times = False; env.export_human_actions = False
        ex = VecExplorer(env, env.robot, gamma=0.9, policy=object())
        outs.append((ex.run_k_episodes(k, "train", action_fn=_goal_seeking, device_scenarios=11), ex.last_records))
    assert outs[0][0] == outs[1][0] and outs[0][1]["returns"] == outs[1][1]["returns"]
    assert len(outs[0][1]["returns"]) == k and len(set(outs[0][1]["infos"])) > 1
    assert env.case_counter["train"] == k % env.case_size["train"]


@pytest.mark.parametrize("stay", [True, False])
def test_data_collection_side_channels_match_sequential_explorer(stay, tmp_path):
    """raw_memory rows, world-model pairs and SGAN cache files (explorer.py:60-85,112-121): the batched VecExplorer
    (E = 4 envs, k = 10 episodes -> 3 rounds with in-kernel restarts) against the sequential Explorer on the E = 1
    gym view, episode by episode, byte for byte."""
    import torch
    import modelcrowdnav_amd.dropin as dropin
    dropin.install()
    import gym
    from crowd_nav.policy.policy_factory import policy_factory
    from crowd_nav.utils.explorer import Explorer
    from crowd_sim.envs.utils.robot import Robot
    from modelcrowdnav_amd import configs
    from modelcrowdnav_amd.rollout import VecExplorer
    from modelcrowdnav_amd.utils import realdata as RD
    device = torch.device("cuda:0")
    k, E = 10, 4
    seq_dir, vec_dir = tmp_path / "seq", tmp_path / "vec"
    seq_dir.mkdir(); vec_dir.mkdir()
    # ---- sequential
    env_config = configs.env_config()
    policy = policy_factory["orca"](); policy.configure(env_config)
    policy.multiagent_training = True                      # 5 humans in 'val' too (crowd_sim.py:275-281)
    env = gym.make("CrowdSim-v0"); env.configure(env_config)
    robot = Robot(env_config, "robot"); robot.set_policy(policy); env.set_robot(robot)
    policy.set_phase("val"); policy.set_device(device); policy.set_env(env)
    robot.policy.safety_space = 0
    ex = Explorer(env, robot, device, None, gamma=0.9)
    ex.batched = False                       # the sequential E = 1 loop itself (k > 1 would go to the batched explorer)
    ex.raw_memory, ex.rawob = [], []
    out = ex.run_k_episodes(k, "val", stay=stay, update_raw_ob=True, cacheFile=str(seq_dir))
    # ---- batched
    venv = H.make_vec_env(E, 5)
    venv.track_human_times = False; venv.export_human_actions = False
    orca = policy_factory["orca"](); orca.multiagent_training = True; orca.safety_space = 0
    venv.robot.set_policy(orca)
    vex = VecExplorer(venv, venv.robot, gamma=0.9, policy=orca)
    vex.raw_memory, vex.rawob = [], []
    vout = vex.run_k_episodes(k, "val", stay=stay, update_raw_ob=True, cacheFile=str(vec_dir))
    assert vout[1:] == out[1:] and abs(vout[0] - out[0]) < 1e-12
    assert len(vex.raw_memory) == len(ex.raw_memory) and len(vex.rawob) == len(ex.rawob) and len(ex.rawob) > 0
    for got, want in zip(vex.raw_memory, ex.raw_memory):
        ob = np.array([[h.px, h.py, h.vx, h.vy, h.radius] for h in want[0]])
        assert np.array_equal(got[0], ob) and got[1] == want[1] and got[2] == want[2] and got[3] == want[3].code
    for got, want in zip(vex.rawob, ex.rawob):
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])
    files = sorted(os.listdir(seq_dir), key=lambda f: int(f.split(".")[0]))
    assert files == ["%d.txt" % (i + 1) for i in range(k)] and sorted(os.listdir(vec_dir)) == sorted(files)
    for f in files:
        assert open(vec_dir / f).read() == open(seq_dir / f).read(), f
    # the cache is what the SGAN side reads back (sgan/sdata/trajectories.py:39-50)
    rows = RD.read_sgan_cache(str(vec_dir / "1.txt"))
    assert rows.shape[1] == 4 and rows[0, 0] == 10.0 and set(rows[:, 1]) == {0.0, 1.0, 2.0, 3.0, 4.0}
    # and the rows feed the batched DataGen
    from modelcrowdnav_amd.utils.datagen import VecDataGen
    import types
    dg = VecDataGen(None, venv.robot, types.SimpleNamespace(device=device), types.SimpleNamespace(gamma=0.9))
    dg.raw_memory = vex.raw_memory
    assert dg.count() == k and dg.load_real_episodes()["obs"].shape[2] == 5
