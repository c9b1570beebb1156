"""Name -> policy constructor table (reference: crowd_sim/envs/policy/policy_factory.py:1-12)."""
from .linear import Linear
from .orca import ORCA

policy_factory = {"linear": Linear, "orca": ORCA, "none": lambda: None}
