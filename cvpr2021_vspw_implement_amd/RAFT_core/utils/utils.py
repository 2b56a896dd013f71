"""RAFT_core/utils/utils.py:7-25,66-69: input padding to multiples of 8 and the pixel-coordinate grid (plumbing)."""
import torch
import torch.nn.functional as F


class InputPadder:
    """Pads images such that dimensions are divisible by 8 (zeros: the reference's replicate mode is commented out)."""

    def __init__(self, dims, mode="sintel"):
        self.ht, self.wd = dims[-2:]
        pad_ht = (((self.ht // 8) + 1) * 8 - self.ht) % 8
        pad_wd = (((self.wd // 8) + 1) * 8 - self.wd) % 8
        if mode == "sintel":
            self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, pad_ht // 2, pad_ht - pad_ht // 2]
        else:
            self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, 0, pad_ht]

    def pad(self, x):
        return F.pad(x, self._pad, mode="constant")

    def unpad(self, x):
        ht, wd = x.shape[-2:]
        c = [self._pad[2], ht - self._pad[3], self._pad[0], wd - self._pad[1]]
        return x[:, :, c[0]:c[1], c[2]:c[3]]


def coords_grid(batch, ht, wd):
    ys, xs = torch.meshgrid(torch.arange(ht), torch.arange(wd), indexing="ij")
    return torch.stack([xs, ys], dim=0).float()[None].repeat(batch, 1, 1, 1)
