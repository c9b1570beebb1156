"""Robot agent (reference: crowd_sim/envs/utils/robot.py:5-14)."""
from .agent import Agent
from .state import JointState


class Robot(Agent):
    def act(self, ob):
        if self.policy is None:
            raise AttributeError("Policy attribute has to be set!")
        return self.policy.predict(JointState(self.get_full_state(), ob))
