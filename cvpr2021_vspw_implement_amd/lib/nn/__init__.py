"""`from lib.nn import user_scattered_collate, async_copy_to` (test_clip2.py:17; reference
lib/nn/parallel/data_parallel.py:13-25,65-66)."""
import collections.abc

import torch


def user_scattered_collate(batch):
    """DataLoader collate_fn that keeps the list of per-sample items as it is."""
    return batch


def async_copy_to(obj, dev, main_stream=None):
    """Tensors (inside dicts / sequences, recursively) -> device `dev` without blocking the host; the copies are
    registered with `main_stream` so the allocator does not recycle them under it.  Anything else passes through."""
    if torch.is_tensor(obj):
        out = obj.cuda(dev, non_blocking=True)
        if main_stream is not None:
            out.record_stream(main_stream)
        return out
    if isinstance(obj, collections.abc.Mapping):
        return {k: async_copy_to(v, dev, main_stream) for k, v in obj.items()}
    if isinstance(obj, collections.abc.Sequence) and not isinstance(obj, (str, bytes)):
        return [async_copy_to(v, dev, main_stream) for v in obj]
    return obj


class UserScatteredDataParallel(object):
    """Single-process multi-GPU wrapper of the reference (data_parallel.py:53-62): replaced by one process per GPU
    (cvpr2021_vspw_implement_amd.distributed)."""

    def __init__(self, *a, **k):
        raise NotImplementedError("UserScatteredDataParallel: this package runs one process per GPU "
                                  "(distributed.DataParallelOverRCCL)")
