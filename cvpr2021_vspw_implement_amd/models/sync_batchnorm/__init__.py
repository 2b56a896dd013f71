"""Name-compatible stand-in for reference models/sync_batchnorm/: the BatchNorm classes run on the HIP kernels and
exchange statistics over torch.distributed (one process per GPU) instead of DataParallel threads."""
from ...nn import SynchronizedBatchNorm1d, SynchronizedBatchNorm2d, SynchronizedBatchNorm3d  # noqa: F401
from .replicate import DataParallelWithCallback, patch_replication_callback  # noqa: F401
