"""`from lib.utils import as_numpy` (test_clip2.py:18; reference lib/utils/th.py:18-28)."""
import collections.abc

import numpy as np
import torch


def as_numpy(obj):
    """Tensors -> numpy arrays on the host, recursively through sequences and mappings; other leaves via np.array."""
    if torch.is_tensor(obj):
        return obj.detach().cpu().numpy()
    if isinstance(obj, collections.abc.Mapping):
        return {k: as_numpy(v) for k, v in obj.items()}
    if isinstance(obj, collections.abc.Sequence) and not isinstance(obj, (str, bytes)):
        return [as_numpy(v) for v in obj]
    return np.array(obj)
