"""Callers on either side of the rollout hot path: replay memory, value targets, trainer."""
