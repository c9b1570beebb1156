"""ModelCrowdSim's own host logic and E = 1 surface against g14_model_env.npz = the real reference's ModelCrowdSim
(crowd_sim/envs/model_crowd_sim.py:94-232,268-345,347-441) driven as its callers drive it."""
import os

import numpy as np
import pytest

INFO = {0: "Nothing", 1: "Danger", 2: "ReachGoal", 3: "Collision", 4: "Timeout"}


def _hum_to_scen(hum):        # fixture rows px,py,vx,vy,radius,gx,gy,v_pref,theta -> scenarios.py's column order
    return np.stack([hum[:, 0], hum[:, 1], hum[:, 5], hum[:, 6], hum[:, 2], hum[:, 3], hum[:, 8], hum[:, 4], hum[:, 7]], 1)


def test_model_env_generators_bitexact(golden_dir):
    """The generators of ModelCrowdSim draw an initial velocity towards (-px, -py) (:183-192,225) and reset() does not
    reseed numpy (:296 is commented out): with the caller's seed, three consecutive resets continue one stream."""
    from modelcrowdnav_amd.envs import scenarios as S
    g = np.load(os.path.join(golden_dir, "g14_model_env.npz"))
    n_cases = len(g["reset_meta_rep"])
    assert n_cases == 216
    moving = 0
    for k in range(n_cases):
        multi, rule, N = bool(g["reset_meta_multiagent"][k]), str(g["reset_meta_rule"][k]), int(g["reset_meta_N"][k])
        rnd, phase, rep = bool(g["reset_meta_randomize"][k]), str(g["reset_meta_phase"][k]), int(g["reset_meta_rep"][k])
        if rep == 0:
            np.random.seed(int(g["reset_meta_seed"][k]))
        spec = S.ScenarioSpec(randomize_attributes=rnd, init_velocity=True)
        if phase == "test":
            hn, r = N, rule
        else:
            hn, r = (N if multi else 1), (rule if multi else "circle_crossing")
        sc = S.generate(spec, np.random, hn, r)
        ref = _hum_to_scen(g["reset_hum_%d" % k])
        assert sc.shape == ref.shape and np.array_equal(sc, ref), (k, rule, phase, rep)
        assert int(g["reset_meta_counter_after"][k]) == 8 + rep          # test_case 7, then the counter runs on
        moving += int(np.any(ref[:, 4:6] != 0))
    assert moving > 150                                   # initial velocities are really in the fixture


@pytest.mark.gpu
def test_e1_reset_set_current_state_and_world_model_episodes_match_reference(golden_dir):
    """The E = 1 drop-in class: reset(-1), set_current_state (goals 0, theta 0, robot from robot_info, :339-345) exact;
    episodes whose humans are moved by an MlpWorld module through step() and onestep_lookahead(): masks / info codes
    exact, floats to 2e-6 (the float32 module runs on the GPU here, on the CPU in the reference; 97 steps accumulate)."""
    import torch
    from modelcrowdnav_amd import configs
    from modelcrowdnav_amd.envs import ModelCrowdSim
    from modelcrowdnav_amd.envs.policy.policy_factory import policy_factory
    from modelcrowdnav_amd.envs.utils.action import ActionXY
    from modelcrowdnav_amd.envs.utils.robot import Robot
    from modelcrowdnav_amd.envs.utils.state import FullState, ObservableState
    from modelcrowdnav_amd.policy.world_model import MlpWorld
    g = np.load(os.path.join(golden_dir, "g14_model_env.npz"))

    def make(N, rule="circle_crossing"):
        cfg = configs.env_config(**{"sim.human_num": N, "sim.train_val_sim": rule, "sim.test_sim": rule})
        env = ModelCrowdSim()
        env.configure(cfg)
        robot = Robot(cfg, "robot")
        robot.set_policy(policy_factory["orca"]())
        env.set_robot(robot)
        return env

    def rows(env):
        r = env.robot
        rob = np.array([r.px, r.py, r.vx, r.vy, r.radius, r.gx, r.gy, r.v_pref, r.theta], np.float64)
        hum = np.array([[h.px, h.py, h.vx, h.vy, h.radius, h.gx, h.gy, h.v_pref, h.theta] for h in env.humans], np.float64)
        return rob, hum.reshape(-1, 9)

    env = make(5)
    assert env.case_size["train"] == int(g["case_size_train"])
    env.reset("test", -1)
    rob, hum = rows(env)
    assert np.array_equal(rob, g["debug_rob"]) and np.array_equal(hum, g["debug_hum"])
    for c in range(4):
        obs = [ObservableState(*r) for r in g["scs_obs_%d" % c]]
        i4 = g["scs_info_%d" % c]
        info = None if i4.size == 0 else FullState(i4[0], i4[1], 0.3, -0.2, 0.3, i4[2], i4[3], 1.0, 0.7)
        env.set_current_state(obs, info, phase=("train", "val", "test", "train")[c])
        rob, hum = rows(env)
        assert np.array_equal(rob, g["scs_rob_%d" % c]), c
        assert np.array_equal(hum, g["scs_hum_%d" % c]), c
        assert env.global_time == float(g["scs_global_time_%d" % c])
    seen = set()
    e = 0
    while "epi%d_seed" % e in g.files:
        hum0 = g["epi%d_hum0" % e]
        N = hum0.shape[0]
        env = make(N, str(g["epi%d_rule" % e]))
        world = MlpWorld(N)
        pref = "epi%d_world__" % e
        world.load_state_dict({k[len(pref):].replace("__", "."): torch.from_numpy(g[k]) for k in g.files if k.startswith(pref)})
        world.eval().to("cuda:0")
        env.sim_world = world
        np.random.seed(int(g["epi%d_seed" % e]))
        env.reset("test")
        rob, hum = rows(env)
        assert np.array_equal(rob, g["epi%d_rob0" % e]) and np.array_equal(hum, hum0), e
        acts = g["epi%d_act" % e]
        tol = 2e-6
        for t in range(acts.shape[0]):
            a = ActionXY(float(acts[t, 0]), float(acts[t, 1]))
            ob, r, d, info = env.onestep_lookahead(a)
            look = np.array([[o.px, o.py, o.vx, o.vy, o.radius] for o in ob])
            np.testing.assert_allclose(look, g["epi%d_look_obs" % e][t], rtol=0, atol=tol)
            assert bool(d) == bool(g["epi%d_look_done" % e][t]) and type(info).__name__ == INFO[int(g["epi%d_look_info" % e][t])]
            assert abs(r - g["epi%d_look_reward" % e][t]) <= tol
            ob, r, d, info = env.step(a)
            got = np.array([[o.px, o.py, o.vx, o.vy, o.radius] for o in ob])
            np.testing.assert_allclose(got, g["epi%d_obs" % e][t], rtol=0, atol=tol)
            assert bool(d) == bool(g["epi%d_done" % e][t]), (e, t)
            assert type(info).__name__ == INFO[int(g["epi%d_info" % e][t])], (e, t)
            assert abs(r - g["epi%d_reward" % e][t]) <= tol
            rob, hum = rows(env)
            np.testing.assert_allclose(rob, g["epi%d_rob" % e][t], rtol=0, atol=1e-12)
            np.testing.assert_allclose(hum, g["epi%d_hum" % e][t], rtol=0, atol=tol)
            assert abs(env.global_time - g["epi%d_time" % e][t]) < 1e-12
            seen.add(type(info).__name__)
        assert d, e                                     # the recorded episode ended here too
        e += 1
    assert e == 6 and seen >= {"Nothing", "Danger", "ReachGoal", "Collision", "Timeout"}, seen
