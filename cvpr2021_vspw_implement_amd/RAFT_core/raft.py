"""RAFT_core/raft.py:26-127 -> the HIP-backed frozen flow network."""
from ..models.raft import RAFT  # noqa: F401
