"""`from models.td4_psp.td4_psp import td4_psp` (train_clip2.py:18, :268; test_clip2.py:15, :215) - TDNet's
distributed sub-networks are not on the hot path this package rebuilds; constructing one raises."""
from ..models import _stub

td4_psp = _stub("td4_psp")
