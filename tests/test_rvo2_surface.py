"""modelcrowdnav_amd.rvo2: the reference's native boundary as a module -- host-side behaviour (no GPU)."""
import numpy as np
import pytest


def _sim():
    from modelcrowdnav_amd import rvo2
    return rvo2.PyRVOSimulator(0.25, 10, 10, 5, 5, 0.3, 1)


def test_surface_has_every_call_the_reference_makes():
    """orca.py:95-129, crowd_sim.py:231-255: constructor with 7 positional arguments, addAgent with 8."""
    sim = _sim()
    for name in ("addAgent", "setAgentPosition", "setAgentVelocity", "setAgentPrefVelocity", "doStep",
                 "getAgentVelocity", "getAgentPosition", "getNumAgents"):
        assert callable(getattr(sim, name)), name
    params = (10, 10, 5, 5)
    assert sim.addAgent((1.0, 2.0), *params, 0.31, 1.0, (0.1, -0.2)) == 0
    assert sim.addAgent((3.0, 4.0), *params, 0.31, 1.0, (0.0, 0.0)) == 1
    assert sim.addAgent((0.5, 0.5)) == 2                                    # constructor defaults
    assert sim.getNumAgents() == 3
    assert sim.getAgentRadius(2) == pytest.approx(0.3) and sim.getAgentMaxSpeed(2) == 1.0


def test_values_pass_through_float32_like_the_cython_layer():
    sim = _sim()
    sim.addAgent((0.1, 0.2), 10, 10, 5, 5, 0.3, 1, (0.3, 0.7))
    assert sim.getAgentPosition(0) == (float(np.float32(0.1)), float(np.float32(0.2)))
    assert sim.getAgentVelocity(0) == (float(np.float32(0.3)), float(np.float32(0.7)))
    sim.setAgentPosition(0, (1 / 3, 2 / 3))
    sim.setAgentVelocity(0, (-1 / 3, 1 / 7))
    sim.setAgentPrefVelocity(0, (5.0, -2.5))
    assert sim.getAgentPosition(0) == (float(np.float32(1 / 3)), float(np.float32(2 / 3)))
    assert sim.getAgentVelocity(0) == (float(np.float32(-1 / 3)), float(np.float32(1 / 7)))
    assert sim.getAgentPrefVelocity(0) == (5.0, -2.5)
    assert isinstance(sim.getAgentPosition(0)[0], float)


def test_rejects_what_the_device_solver_does_not_cover():
    sim = _sim()
    with pytest.raises(ValueError):
        sim.addAgent((0, 0), 10, 11, 5, 5, 0.3, 1, (0, 0))                   # more than MCN_MAX_LINES neighbours
    with pytest.raises(ValueError):
        sim.addAgent((0, 0), 10, 10, 0, 5, 0.3, 1, (0, 0))
    with pytest.raises(NotImplementedError):
        sim.addObstacle([(0, 0), (1, 0), (1, 1)])
    with pytest.raises(NotImplementedError):
        sim.processObstacles()


def test_candidate_lists_keep_agent_order_and_the_nearest_32():
    from modelcrowdnav_amd import rvo2
    sim = rvo2.PyRVOSimulator(0.25, 10, 10, 5, 5, 0.3, 1)
    rng = np.random.RandomState(0)
    pos = rng.uniform(-6, 6, (5, 2)).astype(np.float32)
    idx, m = sim._candidates(pos)
    assert m == 4 and [list(r) for r in idx] == [[j for j in range(5) if j != i] for i in range(5)]
    pos = rng.uniform(-6, 6, (50, 2)).astype(np.float32)
    idx, m = sim._candidates(pos)
    assert m == 32
    for i in range(50):
        d = np.hypot(*(pos - pos[i]).T.astype(np.float64))
        d[i] = np.inf
        want = np.sort(np.argsort(d, kind="stable")[:32])
        assert list(idx[i]) == list(want) and i not in idx[i]


def test_do_step_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    sim = _sim()
    sim.addAgent((0, 0), 10, 10, 5, 5, 0.3, 1, (0, 0))
    sim.addAgent((1, 0), 10, 10, 5, 5, 0.3, 1, (0, 0))
    with pytest.raises(RuntimeError):
        sim.doStep()


def test_install_rvo2_registers_the_module(monkeypatch):
    import sys
    from modelcrowdnav_amd import dropin, rvo2
    monkeypatch.delitem(sys.modules, "rvo2", raising=False)
    assert dropin.install_rvo2() is rvo2
    import importlib
    assert importlib.import_module("rvo2").PyRVOSimulator is rvo2.PyRVOSimulator
    monkeypatch.setitem(sys.modules, "rvo2", object())                      # somebody else's rvo2 stays ...
    assert dropin.install_rvo2() is not rvo2
    assert dropin.install_rvo2(force=True) is rvo2                          # ... unless forced
