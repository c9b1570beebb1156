"""modelcrowdnav_amd.rvo2.PyRVOSimulator (doStep = mcn_orca_batch) against the oracle's solver driven through the same
call sequence (tests/golden_tools/refshim.py's rvo2 stand-in = oracle/mcn_oracle.c): float32, bit-exact.
ORCA parity vs the real rvo2 is unpinned; this pins the module's plumbing and the kernel == oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(dt=0.25, defaults=(10, 10, 5, 5, 0.3, 1)):
    from modelcrowdnav_amd import rvo2
    from tests.golden_tools.refshim import _PyRVOSimulator
    return rvo2.PyRVOSimulator(dt, *defaults), _PyRVOSimulator(dt, *defaults)


def _scene(rng, n, spread):
    pos = rng.uniform(-spread, spread, (n, 2))
    goal = -pos + rng.normal(0, 0.3, (n, 2))
    vel = rng.uniform(-1, 1, (n, 2))
    return pos, vel, goal


@pytest.mark.parametrize("n,spread", [(2, 1.0), (6, 2.0), (6, 0.8), (11, 2.5), (14, 2.0), (33, 4.0), (48, 5.0)])
def test_do_step_matches_oracle_simulator(n, spread):
    """Crowded (infeasible LPs, overlapping discs at spread 0.8), more agents than maxNeighbors (14), more than the 32
    candidates one launch takes (48): positions and velocities after every doStep are the same float32 values."""
    rng = np.random.RandomState(100 + n)
    pos, vel, goal = _scene(rng, n, spread)
    a, b = _pair()
    for sim in (a, b):
        for i in range(n):
            assert sim.addAgent(tuple(pos[i]), 10, 10, 5, 5, 0.3 + 0.01 * (i % 3), 1.0 + 0.1 * (i % 2), tuple(vel[i])) == i
    for step in range(12):
        for sim in (a, b):
            for i in range(n):
                p = np.array(sim.getAgentPosition(i))
                pv = goal[i] - p
                if np.linalg.norm(pv) > 1:
                    pv = pv / np.linalg.norm(pv)
                sim.setAgentPrefVelocity(i, tuple(pv))
            sim.doStep()
        for i in range(n):
            assert a.getAgentVelocity(i) == b.getAgentVelocity(i), (step, i)
            assert a.getAgentPosition(i) == b.getAgentPosition(i), (step, i)
    assert any(a.getAgentVelocity(i) != tuple(np.float32(vel[i]).astype(float)) for i in range(n))
    assert a.getGlobalTime() == pytest.approx(3.0)


def test_reference_call_sequence_of_orca_predict():
    """orca.py:95-129 as the reference drives it: a fresh simulator per call, self = agent 0 with radius + 0.01 +
    safety_space and v_pref as max speed, the others with max speed 1 and preferred velocity (0, 0), the raw goal
    vector as agent 0's preferred velocity, one doStep, read agent 0."""
    rng = np.random.RandomState(7)
    for case in range(20):
        n = int(rng.randint(2, 8))
        pos, vel, goal = _scene(rng, n, 2.0)
        a, b = _pair()
        out = []
        for sim in (a, b):
            params = (10, 10, 5, 5)
            sim.addAgent(tuple(pos[0]), *params, 0.3 + 0.01, 1.0, tuple(vel[0]))
            for i in range(1, n):
                sim.addAgent(tuple(pos[i]), *params, 0.3 + 0.01, 1, tuple(vel[i]))
            sim.setAgentPrefVelocity(0, tuple(goal[0] - pos[0]))
            for i in range(1, n):
                sim.setAgentPrefVelocity(i, (0, 0))
            sim.doStep()
            out.append(sim.getAgentVelocity(0))
        assert out[0] == out[1], case
        assert np.hypot(*out[0]) <= 1.0 + 1e-6


def test_agents_with_their_own_solver_parameters():
    """addAgent's per-agent neighborDist / maxNeighbors / timeHorizon: one launch per distinct triple."""
    rng = np.random.RandomState(3)
    n = 9
    pos, vel, goal = _scene(rng, n, 2.0)
    a, b = _pair()
    prm = [(10, 10, 5), (3.0, 4, 2.0), (10, 2, 5)]
    for sim in (a, b):
        for i in range(n):
            nd, mn, th = prm[i % 3]
            sim.addAgent(tuple(pos[i]), nd, mn, th, 5, 0.3, 1.0, tuple(vel[i]))
            sim.setAgentPrefVelocity(i, tuple(goal[i] - pos[i]))
    for step in range(5):
        a.doStep(); b.doStep()
        for i in range(n):
            assert a.getAgentVelocity(i) == b.getAgentVelocity(i), (step, i)
            assert a.getAgentPosition(i) == b.getAgentPosition(i), (step, i)
