"""`from config import cfg` of the reference drivers (train_clip2.py:14, test_clip2.py:21)."""
from .defaults import _C as cfg  # noqa: F401
from .defaults import CfgNode  # noqa: F401
