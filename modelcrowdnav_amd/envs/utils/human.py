"""Human agent (reference: crowd_sim/envs/utils/human.py:5-17)."""
from .agent import Agent
from .state import JointState


class Human(Agent):
    def act(self, ob):
        # full own state + everybody else's observable state -> the human's policy
        return self.policy.predict(JointState(self.get_full_state(), ob))
