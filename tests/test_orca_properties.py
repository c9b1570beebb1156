"""CPU: pin the ORCA restatement without rvo2 (ORCA parity vs rvo2 is UNPINNED -- the module is
absent from the reference tree and from this image).

What is checked instead, on thousands of random configurations:
  * feasibility  -- the returned velocity satisfies every half-plane (1e-4) and the speed disc
  * optimality   -- when the 2-D LP is feasible, no candidate of an exhaustive active-set
                    enumeration (pref itself, projections on each line / the disc, all line-line
                    and line-disc intersections) is feasible and closer to the preferred velocity
  * fallback     -- the infeasible case (linearProgram3) and the half-plane construction itself are pinned
                    against independent computations in tests/test_orca_pins.py
  * analytic     -- no neighbours / far neighbours -> clip(pref); mirrored head-on pair ->
                    mirrored velocities; half-plane geometry of a single static neighbour
"""
import itertools

import numpy as np

from oracle import cport

EPS = 2e-4


def _viol(lines, v):
    # det(dir, point - v) > 0 means violated; return the max signed violation
    if len(lines) == 0:
        return -np.inf
    p, d = lines[:, 0:2], lines[:, 2:4]
    return np.max(d[:, 0] * (p[:, 1] - v[1]) - d[:, 1] * (p[:, 0] - v[0]))


def _candidates(lines, pref, ms):
    c = [np.array(pref, float)]
    n = np.linalg.norm(pref)
    if n > 0:
        c.append(np.array(pref) / n * ms)
    for ln in lines:
        p, d = ln[0:2], ln[2:4]
        c.append(p + np.dot(pref - p, d) * d)                     # projection on the line
        b, cc = np.dot(p, d), np.dot(p, p) - ms * ms                # line-disc intersections
        disc = b * b - cc
        if disc >= 0:
            for t in (-b - np.sqrt(disc), -b + np.sqrt(disc)):
                c.append(p + t * d)
    for a, b in itertools.combinations(lines, 2):                   # line-line intersections
        den = a[2] * b[3] - a[3] * b[2]
        if abs(den) > 1e-9:
            t = (b[2] * (a[1] - b[1]) - b[3] * (a[0] - b[0])) / den
            c.append(a[0:2] + t * a[2:4])
    return c


def _random_case(rng, n_other, crowded):
    pos = rng.uniform(-1, 1, 2)
    vel = rng.uniform(-1, 1, 2)
    span = 1.2 if crowded else 5.0
    opos = pos + rng.uniform(-span, span, (n_other, 2))
    ovel = rng.uniform(-1, 1, (n_other, 2))
    orad = rng.uniform(0.31, 0.51, n_other)
    pref = rng.uniform(-6, 6, 2) if rng.uniform() < 0.7 else rng.uniform(-0.5, 0.5, 2)
    return pos, vel, 0.31, rng.uniform(0.5, 1.5), pref, opos, ovel, orad


def test_feasible_and_optimal_against_active_set_enumeration():
    rng = np.random.RandomState(42)
    n_feasible = n_infeasible = 0
    for it in range(1200):
        n_other = int(rng.randint(1, 10))
        pos, vel, rad, ms, pref, opos, ovel, orad = _random_case(rng, n_other, crowded=(it % 3 == 0))
        v = np.array(cport.orca_agent(pos, vel, rad, ms, pref, opos, ovel, orad), float)
        lines = cport.orca_lines(pos, vel, rad, opos, ovel, orad).astype(float)
        cands = [c for c in _candidates(lines, pref.astype(np.float32).astype(float), ms)
                 if np.linalg.norm(c) <= ms + EPS and _viol(lines, c) <= EPS]
        if cands:
            n_feasible += 1
            best = min(np.linalg.norm(c - pref) for c in cands)
            assert np.linalg.norm(v) <= ms * (1 + 1e-4) + 1e-5
            assert _viol(lines, v) <= EPS, (it, _viol(lines, v))
            assert np.linalg.norm(v - pref) <= best + 5e-4, (it, np.linalg.norm(v - pref), best)
        else:
            n_infeasible += 1      # the 3-D LP: pinned against SLSQP in tests/test_orca_pins.py
    assert n_feasible > 600 and n_infeasible > 8, (n_feasible, n_infeasible)


def test_analytic_cases():
    # no neighbours: preferred velocity clipped to the speed disc (orca.py:113 passes the raw goal vector)
    vx, vy = cport.orca_agent((0, 0), (0, 0), 0.31, 1.0, (3.0, 4.0), np.zeros((0, 2)), np.zeros((0, 2)), np.zeros(0))
    assert abs(vx - 0.6) < 1e-6 and abs(vy - 0.8) < 1e-6
    vx, vy = cport.orca_agent((0, 0), (0, 0), 0.31, 1.0, (0.3, -0.2), np.zeros((0, 2)), np.zeros((0, 2)), np.zeros(0))
    assert (vx, vy) == (np.float32(0.3), np.float32(-0.2))
    # neighbour beyond neighborDist is ignored even if it sits on the path
    vx, vy = cport.orca_agent((0, 0), (1, 0), 0.31, 1.0, (5.0, 0.0), [[10.5, 0.0]], [[-1, 0]], [0.31])
    assert (vx, vy) == (np.float32(1.0), np.float32(0.0))
    # mirrored head-on pair: velocities are mirror images across the x axis midpoint
    a = cport.orca_agent((-1.0, 0.01), (1, 0), 0.31, 1.0, (4.0, 0.0), [[1.0, -0.01]], [[-1, 0]], [0.31])
    b = cport.orca_agent((1.0, -0.01), (-1, 0), 0.31, 1.0, (-4.0, 0.0), [[-1.0, 0.01]], [[1, 0]], [0.31])
    assert abs(a[0] + b[0]) < 1e-6 and abs(a[1] + b[1]) < 1e-6
    assert a[1] > 0        # each dodges to its own side
    # single static neighbour dead ahead, agent at rest: line direction is a unit vector
    ln = cport.orca_lines((0, 0), (0, 0), 0.31, [[2.0, 0.0]], [[0, 0]], [0.31])
    assert ln.shape == (1, 4) and abs(np.hypot(ln[0, 2], ln[0, 3]) - 1) < 1e-6


def test_neighbour_cap_and_order():
    """maxNeighbors keeps the nearest ones; equal distances keep insertion order."""
    rng = np.random.RandomState(1)
    opos = rng.uniform(-3, 3, (14, 2)); ovel = rng.uniform(-1, 1, (14, 2)); orad = np.full(14, 0.31)
    ln = cport.orca_lines((0, 0), (0.2, 0.1), 0.31, opos, ovel, orad, max_neighbors=10)
    assert ln.shape[0] == 10
    near = np.argsort(np.sum(opos.astype(np.float32) ** 2, 1), kind="stable")[:10]
    ln2 = cport.orca_lines((0, 0), (0.2, 0.1), 0.31, opos[near], ovel[near], orad[near], max_neighbors=10)
    assert np.array_equal(ln, ln2)
