"""Robot policy table (reference: crowd_nav/policy/policy_factory.py:1-8).  'cadrl' and 'lstm_rl' are
outside the path this build covers."""
from ..envs.policy.policy_factory import policy_factory
from .sarl import SARL

policy_factory["sarl"] = SARL
