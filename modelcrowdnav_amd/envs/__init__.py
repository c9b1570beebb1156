"""Environment package (reference: crowd_sim/envs/__init__.py:1-2)."""
from .crowd_sim import CrowdSim, VecCrowdSim          # noqa: F401
from .model_crowd_sim import ModelCrowdSim, VecModelCrowdSim      # noqa: F401
