"""Agent state value types (reference: crowd_sim/envs/utils/state.py:1-55).

Field order is the contract: FullState + ObservableState concatenates to the 14-column joint
row consumed by CADRL.rotate (cadrl.py:223-224).
"""

_FULL = ("px", "py", "vx", "vy", "radius", "gx", "gy", "v_pref", "theta")
_OBS = ("px", "py", "vx", "vy", "radius")


class _State(object):
    _names = ()

    def as_tuple(self):
        return tuple(getattr(self, n) for n in self._names)

    def __add__(self, other):
        # `a + b` puts b's fields first: other + self-fields (state.py:17-18,36-37)
        return other + self.as_tuple()

    def __str__(self):
        return " ".join(str(v) for v in self.as_tuple())


class ObservableState(_State):
    _names = _OBS

    def __init__(self, px, py, vx, vy, radius):
        self.px, self.py, self.vx, self.vy, self.radius = px, py, vx, vy, radius
        self.position = (px, py)
        self.velocity = (vx, vy)

    def getvalue(self):
        return [self.px, self.py, self.vx, self.vy]

    def getvel(self):
        return [self.vx, self.vy]


class FullState(_State):
    _names = _FULL

    def __init__(self, px, py, vx, vy, radius, gx, gy, v_pref, theta):
        self.px, self.py, self.vx, self.vy, self.radius = px, py, vx, vy, radius
        self.gx, self.gy, self.v_pref, self.theta = gx, gy, v_pref, theta
        self.position = (px, py)
        self.goal_position = (gx, gy)
        self.velocity = (vx, vy)


class JointState(object):
    def __init__(self, self_state, human_states):
        if not isinstance(self_state, FullState):
            raise AssertionError("self_state must be a FullState")
        for s in human_states:
            if not isinstance(s, ObservableState):
                raise AssertionError("human_states must be ObservableState")
        self.self_state = self_state
        self.human_states = human_states
