"""`patch_replication_callback` is called by the reference drivers right after wrapping the model in
nn.DataParallel (train_clip2.py:359-364).  With one process per GPU there is no replication to patch: statistics
are synchronised by ops.set_sync_bn() over RCCL.  The function is kept so the drivers run unchanged."""


def patch_replication_callback(data_parallel):
    return data_parallel


class DataParallelWithCallback(object):
    def __init__(self, *a, **k):
        raise NotImplementedError("single-process DataParallel is replaced by one process per GPU "
                                  "(cvpr2021_vspw_implement_amd.distributed)")
