"""Import-surface mirror of the reference's RAFT_core/utils/utils.py (InputPadder :7-25, coords_grid :66-69) on this
package's own kernels: padding and cropping are ONE gather each (ops.plane_shift, csrc/misc.hip), not ATen copies.  The
product path does not come through here (models/netwarp.py pads with the same helper); TC_cal.py and callers that were
written against the reference's module do."""
import torch

from ... import ops


def margins_to_multiple_of_8(height, width, mode="sintel"):
    """(top, bottom, left, right) zero margins that bring height x width up to multiples of 8: split evenly ('sintel',
    the extra row / column at the bottom / right) or all of the height margin at the bottom (any other mode, the
    reference's KITTI branch).  Zeros, not edge replication: the reference's replicate mode is commented out."""
    extra_h, extra_w = -height % 8, -width % 8
    left = extra_w // 2
    if mode == "sintel":
        top = extra_h // 2
        return top, extra_h - top, left, extra_w - left
    return 0, extra_h, left, extra_w - left


class InputPadder:
    """pad(x): [N, C, H, W] device tensor -> zero-padded to multiples of 8; unpad(y) crops a tensor of the padded size
    (or of any size carrying the same margins) back."""

    def __init__(self, dims, mode="sintel"):
        self.ht, self.wd = int(dims[-2]), int(dims[-1])
        self.top, self.bottom, self.left, self.right = margins_to_multiple_of_8(self.ht, self.wd, mode)
        self._pad = [self.left, self.right, self.top, self.bottom]  # (the reference's attribute, F.pad order)

    @property
    def padded_size(self):
        return self.ht + self.top + self.bottom, self.wd + self.left + self.right

    def pad(self, x):
        return ops.plane_shift(x, self.padded_size, self.top, self.left)

    def unpad(self, x):
        h, w = x.shape[-2:]
        return ops.plane_shift(x, (h - self.top - self.bottom, w - self.left - self.right), -self.top, -self.left)


def coords_grid(batch, ht, wd):
    """[batch, 2, ht, wd] pixel coordinates, channel 0 = x, channel 1 = y (host tensor, like the reference's)."""
    x = torch.arange(wd, dtype=torch.float32).view(1, 1, 1, wd).expand(batch, 1, ht, wd)
    y = torch.arange(ht, dtype=torch.float32).view(1, 1, ht, 1).expand(batch, 1, ht, wd)
    return torch.cat([x, y], dim=1).contiguous()
