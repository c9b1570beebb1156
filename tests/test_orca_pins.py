"""CPU: independent pins of the ORCA restatement (crowd_sim/envs/policy/orca.py:95-129 -> rvo2).

ORCA parity vs rvo2 stays UNPINNED (the module is absent).  What this file pins instead is the oracle against the
*published definitions*, each by an independent float64 computation that shares no formula with oracle/mcn_oracle.c:

  (a) half-plane geometry (van den Berg et al. 2011, sec. 4):  for the pair (A, B) with relPos = p_B - p_A,
      relVel = v_A - v_B, R = r_A + r_B, the velocity obstacle for the window tau is
          VO = union over t in (0, tau] of disc(relPos / t, R / t);
      u is the vector from relVel to the NEAREST point of VO's boundary, n the outward normal there, and A's permitted
      half-plane is { v : (v - (v_A + u / 2)) . n >= 0 }.  Already-colliding pairs use the disc of the window
      timeStep alone (RVO2's convention: get out within one step).
      VO is decomposed here by trigonometry (cone half-angle asin(R / |relPos|), tangent length), not by RVO2's
      leg formulas, and membership is tested straight from the union definition.
  (b) linearProgram3 (sec. 5.3: minimise the maximum penetration inside the speed disc) against
      scipy.optimize.minimize(SLSQP) on the epigraph form; linearProgram2 against SLSQP on the QP.
  (c) both on inputs taken from the benchmark workload itself (BASELINE configs[1]: circle crossing, 5 humans,
      invisible robot), dumped from the oracle's own env loop.
"""
import numpy as np
import pytest
from scipy.optimize import minimize

from oracle import cport

TAU, DT = 5.0, 0.25


# ---------------------------------------------------------------- (a) independent velocity-obstacle geometry
def _in_vo(p, rp, R, tau):
    """Membership from the definition: exists t in (0, tau] with |p - rp / t| < R / t  <=>  min_t |t p - rp| < R."""
    pp = float(np.dot(p, p))
    t = tau if pp == 0.0 else min(max(float(np.dot(p, rp)) / pp, 1e-12), tau)
    return np.linalg.norm(t * p - rp) < R


def _vo_nearest(rp, R, tau, p):
    """Nearest boundary point of the truncated cone to p, its outward normal, the part it lies on, and the gap to
    the runner-up part (ties between parts are measure-zero but float32 can land on either side of one)."""
    L = np.linalg.norm(rp)
    phi, alpha = np.arctan2(rp[1], rp[0]), np.arcsin(R / L)
    c, r, tlen = rp / tau, R / tau, np.sqrt(L * L - R * R) / tau
    out = []
    # the cap: arc of circle(c, r) that faces the origin, +-(pi/2 - alpha) around the direction c -> origin
    w = p - c
    th = np.arctan2(w[1], w[0]) - (phi + np.pi)
    th = (th + np.pi) % (2 * np.pi) - np.pi
    beta = np.pi / 2 - alpha
    th_c = min(max(th, -beta), beta)
    nrm = np.array([np.cos(phi + np.pi + th_c), np.sin(phi + np.pi + th_c)])
    out.append((c + r * nrm, nrm, "arc"))
    # the two legs: rays from the tangent points along the cone's edges
    for sgn, name in ((+1, "left"), (-1, "right")):
        e = np.array([np.cos(phi + sgn * alpha), np.sin(phi + sgn * alpha)])
        t = max(float(np.dot(p, e)), tlen)
        nn = np.array([np.cos(phi + sgn * (alpha + np.pi / 2)), np.sin(phi + sgn * (alpha + np.pi / 2))])
        out.append((t * e, nn, name))
    d = [np.linalg.norm(q - p) for q, _, _ in out]
    k = int(np.argmin(d))
    gap = sorted(d)[1] - d[k]
    return out[k][0], out[k][1], out[k][2], d[k], gap


def _check_line(pos, vel, rad, opos, ovel, orad, line, stats, tol=2e-5):
    """One ORCA half-plane (point, direction) of agent A against neighbour B."""
    pos, vel, opos, ovel = (np.asarray(x, np.float32).astype(np.float64) for x in (pos, vel, opos, ovel))
    R = float(np.float32(rad)) + float(np.float32(orad))
    rp, rv = opos - pos, vel - ovel
    point, d = line[0:2].astype(np.float64), line[2:4].astype(np.float64)
    u = 2.0 * (point - vel)                                  # the line passes through v_A + u / 2 (reciprocity)
    b = rv + u
    n_line = np.array([-d[1], d[0]])                         # permitted side = left of the direction
    assert abs(np.linalg.norm(d) - 1) < 1e-5
    scale = max(1.0, np.linalg.norm(b))
    if np.dot(rp, rp) > R * R:
        q, n, part, dist, gap = _vo_nearest(rp, R, TAU, rv)
        inside = _in_vo(rv, rp, R, TAU)
        stats[part + ("_in" if inside else "_out")] = stats.get(part + ("_in" if inside else "_out"), 0) + 1
        # |u| is the distance to the boundary, and rv + u is ON the boundary by the union definition
        assert abs(np.linalg.norm(u) - dist) <= tol * scale, (part, np.linalg.norm(u), dist)
        eps = 2e-4 * scale
        assert not _in_vo(b + eps * n_line, rp, R, TAU) and _in_vo(b - eps * n_line, rp, R, TAU), part
        if gap > 1e-3:                                       # away from a tie it is THE nearest point, same normal
            assert np.linalg.norm(b - q) <= tol * scale, (part, b, q)
            assert np.dot(n_line, n) >= 1 - 1e-5, (part, n_line, n)
        if np.linalg.norm(u) > 1e-4:
            # u is normal to the boundary: parallel to n (outward when relVel is inside, inward when outside)
            assert abs(np.dot(u, d)) <= 1e-5 * scale
            assert np.dot(u, n_line) * (1 if inside else -1) > 0
    else:
        stats["collide"] = stats.get("collide", 0) + 1
        c, r = rp / DT, R / DT
        w = rv - c
        nw = w / np.linalg.norm(w)
        q = c + r * nw
        assert np.linalg.norm(b - q) <= 1e-5 * max(1.0, r), (b, q)
        assert np.dot(n_line, nw) >= 1 - 1e-5
        assert abs(np.dot(u, d)) <= 1e-5 * max(1.0, r)


def _random_pair(rng, kind):
    pos, vel = rng.uniform(-4, 4, 2), rng.uniform(-1.2, 1.2, 2)
    rad, orad = rng.uniform(0.3, 0.55), rng.uniform(0.3, 0.55)
    R = np.float32(rad) + np.float32(orad)
    ang = rng.uniform(0, 2 * np.pi)
    if kind == "collide":
        dist = rng.uniform(0.05, 0.97) * R
    elif kind == "near":
        dist = R * rng.uniform(1.02, 1.6)
    else:
        dist = rng.uniform(1.0, 9.0)
    opos = pos + dist * np.array([np.cos(ang), np.sin(ang)])
    if kind == "toward":      # a neighbour on a collision course: relVel well inside the cone / beyond the cap
        ovel = -rng.uniform(0.2, 1.2) * np.array([np.cos(ang), np.sin(ang)]) + rng.uniform(-0.3, 0.3, 2)
        vel = rng.uniform(0.2, 1.2) * np.array([np.cos(ang), np.sin(ang)]) + rng.uniform(-0.3, 0.3, 2)
    else:
        ovel = rng.uniform(-1.2, 1.2, 2)
    if kind == "cap":         # relVel inside the rounded cap of the truncated cone (slow approach, long horizon)
        rp = np.float32(opos).astype(float) - np.float32(pos).astype(float)
        beta = np.arccos(min(1.0, R / np.linalg.norm(rp)))
        th = np.arctan2(-rp[1], -rp[0]) + rng.uniform(-0.9, 0.9) * beta
        rv = rp / TAU + rng.uniform(0.2, 0.98) * (R / TAU) * np.array([np.cos(th), np.sin(th)])
        vel = ovel + rv
    return pos, vel, rad, opos, ovel, orad


def test_half_planes_are_the_velocity_obstacle_construction():
    rng = np.random.RandomState(7)
    stats = {}
    kinds = ["any", "toward", "near", "collide", "cap"]
    for it in range(5000):
        pos, vel, rad, opos, ovel, orad = _random_pair(rng, kinds[it % 5])
        ln = cport.orca_lines(pos, vel, rad, [opos], [ovel], [orad])
        assert ln.shape == (1, 4)
        _check_line(pos, vel, rad, opos, ovel, orad, ln[0], stats)
        # reciprocity: B's line against A is the mirror image, u_B = -u_A (each takes half of the avoidance)
        lb = cport.orca_lines(opos, ovel, orad, [pos], [vel], [rad])[0]
        ua = 2 * (ln[0, 0:2].astype(float) - np.float32(vel).astype(float))
        ub = 2 * (lb[0:2].astype(float) - np.float32(ovel).astype(float))
        assert np.allclose(ua, -ub, atol=2e-5 * max(1.0, np.abs(ua).max())), (ua, ub)
        assert np.allclose(ln[0, 2:4], -lb[2:4], atol=1e-5)
    # every branch of the construction was exercised, from both sides of the boundary
    for k in ("arc_in", "arc_out", "left_in", "left_out", "right_in", "right_out", "collide"):
        assert stats.get(k, 0) >= 100, stats
    assert stats["collide"] >= 900


# ---------------------------------------------------------------- (b) the two linear programs against SLSQP
def _pen(lines, v):
    p, d = lines[:, 0:2], lines[:, 2:4]
    return d[:, 0] * (p[:, 1] - v[1]) - d[:, 1] * (p[:, 0] - v[0])       # > 0: violated by that much


def _slsqp_minimax(lines, ms):
    """min z  s.t.  z >= pen_i(v),  |v| <= ms   (epigraph form of RVO2's linearProgram3 objective)."""
    p, d = lines[:, 0:2], lines[:, 2:4]
    A = np.stack([d[:, 1], -d[:, 0]], 1)                                   # pen_i(v) = c_i + A_i . v
    c0 = d[:, 0] * p[:, 1] - d[:, 1] * p[:, 0]
    cons = [{"type": "ineq", "fun": lambda x: x[2] - c0 - A @ x[0:2], "jac": lambda x: np.hstack([-A, np.ones((len(A), 1))])},
            {"type": "ineq", "fun": lambda x: ms * ms - x[0] ** 2 - x[1] ** 2,
             "jac": lambda x: np.array([-2 * x[0], -2 * x[1], 0.0])}]
    best = None
    for v0 in (np.zeros(2), 0.5 * ms * np.array([1.0, 0.0]), 0.5 * ms * np.array([-0.5, 0.8])):
        x0 = np.array([v0[0], v0[1], np.max(c0 + A @ v0)])
        r = minimize(lambda x: x[2], x0, jac=lambda x: np.array([0.0, 0.0, 1.0]), constraints=cons, method="SLSQP",
                     options={"ftol": 1e-14, "maxiter": 300})
        ok = r.success and r.x[0] ** 2 + r.x[1] ** 2 <= ms * ms * (1 + 1e-8)
        if ok and (best is None or r.x[2] < best):
            best = float(np.max(c0 + A @ r.x[0:2]))
    return best


def _slsqp_qp(lines, ms, pref):
    """min |v - pref|^2  s.t. every half-plane and the speed disc (linearProgram2's problem)."""
    p, d = lines[:, 0:2], lines[:, 2:4]
    A = np.stack([d[:, 1], -d[:, 0]], 1)
    c0 = d[:, 0] * p[:, 1] - d[:, 1] * p[:, 0]
    cons = [{"type": "ineq", "fun": lambda x: -(c0 + A @ x), "jac": lambda x: -A},
            {"type": "ineq", "fun": lambda x: ms * ms - x @ x, "jac": lambda x: -2 * x}]
    best = None
    for v0 in (np.zeros(2), pref / max(1.0, np.linalg.norm(pref) / ms)):
        r = minimize(lambda x: (x - pref) @ (x - pref), v0, jac=lambda x: 2 * (x - pref), constraints=cons,
                     method="SLSQP", options={"ftol": 1e-14, "maxiter": 300})
        if r.success and np.max(c0 + A @ r.x) <= 1e-7 and r.x @ r.x <= ms * ms * (1 + 1e-8):
            val = float(np.linalg.norm(r.x - pref))
            best = val if best is None else min(best, val)
    return best


def _check_solution(lines, ms, pref, v, stats, tol=1e-4):
    """The oracle's velocity v for half-planes `lines` against the two published optimisation problems."""
    lines = lines.astype(np.float64)
    v = np.asarray(v, np.float64)
    pref = np.asarray(np.float32(pref), np.float64)
    if len(lines) == 0:
        return
    zstar = _slsqp_minimax(lines, ms)
    if zstar is None:
        stats["slsqp_failed"] = stats.get("slsqp_failed", 0) + 1
        return
    if zstar > 2e-5:                      # no velocity satisfies all half-planes: linearProgram3's territory
        if np.linalg.norm(v) > ms * (1 + 1e-3):
            # float32 cancellation in linearProgram1's disc test on near-parallel projected lines: a property of
            # the published float32 algorithm, reproduced as is; bounded below
            stats["lp3_illcond"] = stats.get("lp3_illcond", 0) + 1
            return
        stats["lp3"] = stats.get("lp3", 0) + 1
        got = float(np.max(_pen(lines, v)))
        assert abs(got - zstar) <= tol, (got, zstar, len(lines))
        stats["lp3_maxgap"] = max(stats.get("lp3_maxgap", 0.0), abs(got - zstar))
    elif zstar < -2e-5:                   # strictly feasible: linearProgram2's territory
        stats["lp2"] = stats.get("lp2", 0) + 1
        assert np.linalg.norm(v) <= ms * (1 + 1e-5) + 1e-6
        assert np.max(_pen(lines, v)) <= 2e-5
        best = _slsqp_qp(lines, ms, pref)
        if best is not None:
            assert abs(np.linalg.norm(v - pref) - best) <= tol, (np.linalg.norm(v - pref), best)
            stats["lp2_maxgap"] = max(stats.get("lp2_maxgap", 0.0), abs(np.linalg.norm(v - pref) - best))
    else:
        stats["borderline"] = stats.get("borderline", 0) + 1


def _crowded_case(rng, n_other):
    pos, vel = rng.uniform(-1, 1, 2), rng.uniform(-1, 1, 2)
    # neighbours packed around the agent and closing in: most of these have an empty feasible region
    ang = rng.uniform(0, 2 * np.pi, n_other)
    dist = rng.uniform(0.63, 1.1, n_other)
    opos = pos + np.stack([dist * np.cos(ang), dist * np.sin(ang)], 1)
    ovel = -rng.uniform(0.3, 1.2, (n_other, 1)) * np.stack([np.cos(ang), np.sin(ang)], 1) + rng.uniform(-0.3, 0.3, (n_other, 2))
    orad = np.full(n_other, 0.31)
    pref = rng.uniform(-6, 6, 2)
    return pos, vel, 0.31, rng.uniform(0.5, 1.5), pref, opos, ovel, orad


def test_linear_program_3_is_the_minimax_optimum():
    rng = np.random.RandomState(11)
    stats = {}
    for it in range(1800):
        n_other = int(rng.randint(3, 10))
        pos, vel, rad, ms, pref, opos, ovel, orad = _crowded_case(rng, n_other)
        v = cport.orca_agent(pos, vel, rad, ms, pref, opos, ovel, orad)
        lines = cport.orca_lines(pos, vel, rad, opos, ovel, orad)
        _check_solution(lines, ms, pref, v, stats)
    assert stats.get("lp3", 0) >= 1000, stats
    assert stats.get("lp3_illcond", 0) <= 0.05 * stats["lp3"], stats
    assert stats.get("slsqp_failed", 0) <= 0.02 * 1800, stats


def test_linear_program_2_is_the_qp_optimum():
    rng = np.random.RandomState(13)
    stats = {}
    for it in range(800):
        n_other = int(rng.randint(1, 10))
        pos, vel = rng.uniform(-1, 1, 2), rng.uniform(-1, 1, 2)
        opos = pos + rng.uniform(-4, 4, (n_other, 2))
        ovel, orad = rng.uniform(-1, 1, (n_other, 2)), rng.uniform(0.31, 0.51, n_other)
        pref = rng.uniform(-6, 6, 2) if it % 3 else rng.uniform(-0.5, 0.5, 2)
        ms = rng.uniform(0.5, 1.5)
        v = cport.orca_agent(pos, vel, 0.31, ms, pref, opos, ovel, orad)
        lines = cport.orca_lines(pos, vel, 0.31, opos, ovel, orad)
        _check_solution(lines, ms, pref, v, stats)
    assert stats.get("lp2", 0) >= 500, stats


# ---------------------------------------------------------------- (c) the benchmark workload's own states
def _bench_workload_agents(n_envs, steps, seed):
    """(agent, neighbours) inputs exactly as the oracle's env loop hands them to the solver on BASELINE configs[1]
    (mcn_oracle.c env step <- orca.py:95-129: radius r + 0.01, raw goal vector as preferred velocity, v_pref as
    the speed limit, invisible robot), at the steps where circle crossing is densest."""
    from modelcrowdnav_amd.envs import scenarios as S
    from modelcrowdnav_amd.policy.cadrl import build_action_space
    from oracle import cpu_replica
    N = 5
    pool = S.scenario_pool(S.ScenarioSpec(), "test", range(n_envs), N, "circle_crossing")
    tab = build_action_space(1.0, "holonomic", 5, 16)[0]
    w = cpu_replica.setup(pool[np.arange(n_envs)], tab, seed)
    st, cfg, rng = w["st"], w["cfg"], w["rng"]
    out = []
    for t in range(max(steps) + 1):
        if t in steps:
            f32 = lambda a: a.astype(np.float32)
            for e in range(n_envs):
                for i in range(N):
                    o = [j for j in range(N) if j != i]
                    out.append(dict(
                        pos=(st.hpx[e, i], st.hpy[e, i]), vel=(st.hvx[e, i], st.hvy[e, i]),
                        rad=np.float32(st.hr[e, i] + 0.01), ms=float(np.float32(st.hvpref[e, i])),
                        pref=(np.float32(st.hgx[e, i] - st.hpx[e, i]), np.float32(st.hgy[e, i] - st.hpy[e, i])),
                        opos=np.stack([st.hpx[e, o], st.hpy[e, o]], 1), ovel=np.stack([st.hvx[e, o], st.hvy[e, o]], 1),
                        orad=f32(st.hr[e, o] + 0.01)))
        a = tab[rng.randint(0, 81, n_envs)]
        o = cport.env_step(cfg, st, np.ascontiguousarray(a[:, 0]), np.ascontiguousarray(a[:, 1]))
        # the solver's outputs of this step, to tie the dump to what the env loop really computed
        if t in steps:
            for k, e_i in enumerate((e, i) for e in range(n_envs) for i in range(N)):
                out[len(out) - n_envs * N + k]["act"] = o["human_act"][e_i[0], e_i[1]]
    return out


@pytest.fixture(scope="module")
def bench_agents():
    return _bench_workload_agents(60, (8, 12, 15, 17, 19, 21, 24), seed=3)


def test_bench_workload_half_planes(bench_agents):
    stats, n_lines = {}, 0
    for a in bench_agents:
        lines = cport.orca_lines(a["pos"], a["vel"], a["rad"], a["opos"], a["ovel"], a["orad"])
        # all four neighbours are inside neighborDist = 10 on this workload; lines come sorted by distance
        order = np.argsort(np.sum((np.float32(a["opos"]) - np.float32(a["pos"])) ** 2, 1), kind="stable")
        assert len(lines) == len(order)
        for ln, j in zip(lines, order):
            _check_line(a["pos"], a["vel"], a["rad"], a["opos"][j], a["ovel"][j], a["orad"][j], ln, stats)
            n_lines += 1
    assert n_lines >= 8000 and sum(v for k, v in stats.items() if k.endswith("_in")) >= 200, (n_lines, stats)


def test_bench_workload_velocities_are_optimal(bench_agents):
    stats = {}
    for a in bench_agents:
        v = cport.orca_agent(a["pos"], a["vel"], a["rad"], a["ms"], a["pref"], a["opos"], a["ovel"], a["orad"])
        assert np.array_equal(np.float64(v), a["act"])          # this IS what the env loop's step produced
        lines = cport.orca_lines(a["pos"], a["vel"], a["rad"], a["opos"], a["ovel"], a["orad"])
        _check_solution(lines, a["ms"], a["pref"], v, stats)
    # the workload reaches both programs: most agents solve the 2-D LP, the dense phase falls through to the 3-D LP
    assert stats.get("lp2", 0) >= 1000 and stats.get("lp3", 0) >= 20, stats
