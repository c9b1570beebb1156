"""modelcrowdnav_amd -- MI355X-native implementation of ModelCrowdNav's data-parallel rollout hot path.

  envs/    CrowdSim / ModelCrowdSim gym surface (E = 1) and VecCrowdSim (E envs resident in HBM)
  policy/  SARL attention value network and its 81-action look-ahead, SGAN world model
  csrc/    hand-written HIP kernels for gfx950 + the C ABI (include/mcn.h)

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); every
hot op is a HIP kernel reached through the C ABI.  There is no CPU fallback.
"""
__version__ = "0.4.0"
