"""Helpers kept from the reference's sgan/utils.py that the rollout path touches."""
import torch


def relative_to_abs(rel_traj, start_pos):
    """sgan/utils.py:85-98: cumulative displacements [T,B,2] + start [B,2] -> absolute [T,B,2]."""
    return torch.cumsum(rel_traj, dim=0) + start_pos.unsqueeze(0)
