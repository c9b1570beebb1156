"""CPU: the oracle (C restatement) and the host logic against fixtures recorded from the real
reference (tests/golden_tools/gen_golden.py).  These pin the oracle before it is trusted as the GPU checker."""
import os

import numpy as np
import pytest

from oracle import cport


def _load_state(g, i):
    N = int(g["N"][i])
    st = cport.EnvState(1, N)
    h, r = g["hum_in"][i][:N], g["rob_in"][i]
    st.hpx[0], st.hpy[0], st.hvx[0], st.hvy[0], st.hr[0] = h[:, 0], h[:, 1], h[:, 2], h[:, 3], h[:, 4]
    st.hgx[0], st.hgy[0], st.hvpref[0] = h[:, 5], h[:, 6], h[:, 7]
    st.rpx[0], st.rpy[0], st.rvx[0], st.rvy[0], st.rr[0], st.rgx[0], st.rgy[0] = r[0], r[1], r[2], r[3], r[4], r[5], r[6]
    st.gtime[0] = g["time"][i]
    st.rtheta[0] = r[8]
    return st, N


@pytest.mark.parametrize("name,policy,tol", [("g2_step_given", cport.HUMANS_GIVEN, 0.0),
                                             ("g2_step_unicycle", cport.HUMANS_GIVEN, 1e-14),
                                             ("g2_step_linear", cport.HUMANS_LINEAR, 1e-14),
                                             ("g2_step_orca", cport.HUMANS_ORCA, 0.0),
                                             ("g2_step_orca_visible", cport.HUMANS_ORCA, 0.0)])
def test_env_step_oracle_matches_reference(name, policy, tol, golden_dir):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    vis = int(bool(g["robot_visible"]))
    for i in range(len(g["N"])):
        st, N = _load_state(g, i)
        cfg = cport.default_cfg(robot_visible=vis, human_policy=policy, count_hh=int(policy != cport.HUMANS_GIVEN),
                                robot_unicycle=int("unicycle" in name))
        upd = bool(g["update"][i])
        out = cport.env_step(cfg, st, g["act"][i][:1].copy(), g["act"][i][1:2].copy(), update=upd,
                             given_v=g["given_v"][i][:N].copy())
        assert out["done"][0] == g["done"][i] and out["info"][0] == g["info"][i], i
        assert abs(out["reward"][0] - g["reward"][i]) <= tol
        if g["info"][i] == cport.INFO_DANGER:
            assert abs(out["dmin"][0] - g["dmin"][i]) <= tol
        if upd:
            ho, ro = g["hum_out"][i][:N], g["rob_out"][i]
            got = np.stack([st.hpx[0], st.hpy[0], st.hvx[0], st.hvy[0]], 1)
            np.testing.assert_allclose(got, ho[:, 0:4], rtol=0, atol=tol)
            np.testing.assert_allclose([st.rpx[0], st.rpy[0], st.rvx[0], st.rvy[0]], ro[0:4], rtol=0, atol=tol)
            if "unicycle" in name:
                assert abs(st.rtheta[0] - ro[8]) <= tol
            assert st.gtime[0] == g["time_out"][i]
            if policy != cport.HUMANS_GIVEN:
                np.testing.assert_array_equal(st.human_times[0], g["human_times"][i][:N])
        else:
            ob = g["obs"][i][:N]
            got = np.stack([out["nobs_px"][0], out["nobs_py"][0], out["nobs_vx"][0], out["nobs_vy"][0]], 1)
            np.testing.assert_allclose(got, ob[:, 0:4], rtol=0, atol=tol)


def test_point_to_segment_dist_bitexact(golden_dir):
    g = np.load(os.path.join(golden_dir, "g3_p2s.npz"))
    d = np.array([cport.point_to_segment_dist(*a) for a in g["args"]])
    assert np.array_equal(d, g["dist"])
    # known answers quoted in SURVEY.md 8c (G3)
    assert cport.point_to_segment_dist(1, 1, 2, 3, 0, 0) == 1.4142135623730951
    assert cport.point_to_segment_dist(-1, 1, 2, 1, 0, 0) == 1.0


def test_scenarios_bitexact(golden_dir):
    """Host scenario generator == CrowdSim.reset for every rule / phase / N / randomisation."""
    from modelcrowdnav_amd.envs import scenarios as S
    g = np.load(os.path.join(golden_dir, "g1_reset.npz"))
    for k in range(len(g["meta_case"])):
        multi, rule, N = bool(g["meta_multiagent"][k]), str(g["meta_rule"][k]), int(g["meta_N"][k])
        rnd, phase, case = bool(g["meta_randomize"][k]), str(g["meta_phase"][k]), int(g["meta_case"][k])
        spec = S.ScenarioSpec(randomize_attributes=rnd)
        if phase == "test":
            hn, r = N, rule
        else:
            hn, r = (N if multi else 1), (rule if multi else "circle_crossing")
        sc = S.scenario_for_case(spec, phase, case, hn, r)
        hum = g["hum_%d" % k]      # px,py,vx,vy,radius,gx,gy,v_pref,theta
        ref = np.stack([hum[:, 0], hum[:, 1], hum[:, 5], hum[:, 6], hum[:, 2], hum[:, 3], hum[:, 8], hum[:, 4], hum[:, 7]], 1)
        assert sc.shape == ref.shape and np.array_equal(sc, ref), (k, rule, phase, case)
    sc = S.scenario_for_case(S.ScenarioSpec(), "test", 0, 5, "circle_crossing")
    assert sc[0, S.PX] == -2.6625559084662678 and sc[0, S.PY] == -2.837985290326649


def test_action_table_bitexact(golden_dir):
    from modelcrowdnav_amd.policy.cadrl import build_action_space
    g = np.load(os.path.join(golden_dir, "g4_actions.npz"))
    for kin in ("holonomic", "unicycle"):
        for vp in (1.0, 0.7):
            t, s, r = build_action_space(vp, kin)
            assert np.array_equal(t, g["%s_%g" % (kin, vp)])
            assert np.array_equal(np.array(s), g["speeds_%s_%g" % (kin, vp)])
            assert np.array_equal(np.array(r), g["rotations_%s_%g" % (kin, vp)])
    t, s, _ = build_action_space(1.0)
    assert s[0] == 0.12885124808584156 and tuple(t[6]) == (0.11904303084504313, 0.04930923788201555)


def test_pow_half_vs_sqrt_never_flips_the_overlap_test():
    """crowd_sim.py:371 computes the human-human distance as (dx**2 + dy**2)**(1/2) -- CPython float pow, i.e. libm
    pow(x, 0.5), which the oracle keeps; the kernels use the correctly rounded sqrt (documented deviation).  On this libm the two differ by one ulp
    for ~0.05 % of arguments; what the env derives from it is only the sign of `dist - r_i - r_j`, and that sign is the
    same for every argument within 2000 ulps of the touching distance of the shipped radii (exhaustive); for random
    radii it can differ only where |dist - r_i - r_j| is itself below one ulp (a measure-zero boundary; the count is
    only logged by the reference, crowd_sim.py:375-376)."""
    rng = np.random.RandomState(0)
    x = rng.uniform(0, 200, 300000)
    a = np.array([float(v) ** (1 / 2) for v in x])
    assert np.mean(a != np.sqrt(x)) < 2e-3 and np.max(np.abs(a - np.sqrt(x)) / np.sqrt(x)) < 2.3e-16
    for ri, rj in [(0.3, 0.3), (0.3, 0.45), (0.5, 0.31), (0.37, 0.42)]:
        touch = (ri + rj) ** 2
        xs = [touch]
        lo = hi = touch
        for _ in range(2000):
            lo, hi = np.nextafter(lo, 0.0), np.nextafter(hi, 10.0)
            xs += [lo, hi]
        for v in xs:
            assert ((float(v) ** (1 / 2) - ri - rj) < 0) == ((np.sqrt(v) - ri - rj) < 0), (ri, rj, v)
    r1, r2 = rng.uniform(0.3, 0.5, 200000), rng.uniform(0.3, 0.5, 200000)
    xs = (r1 + r2) ** 2 * (1 + rng.uniform(-1e-15, 1e-15, 200000))
    pw = np.array([float(v) ** (1 / 2) for v in xs])
    d_pow, d_sqrt = pw - r1 - r2, np.sqrt(xs) - r1 - r2
    flips = (d_pow < 0) != (d_sqrt < 0)
    assert np.all(np.abs(d_sqrt[flips]) < 4.5e-16)         # only pairs within an ulp of touching can differ
