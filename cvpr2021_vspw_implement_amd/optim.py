"""SGD for the clip heads: torch.optim.SGD semantics, executed by one HIP kernel per parameter.

The reference builds four parameter groups from generators that yield a parameter once per enclosing module
(models/clip_psp.py:99-135), and torch.optim.SGD (a Python loop in the pinned 1.3.1) then applies the momentum
update once per occurrence.  This optimizer keeps that behaviour exactly — group['params'] may contain duplicates —
but folds the k occurrences of a parameter into ONE kernel launch that applies the update k times in registers.
"""
import ctypes
from collections import OrderedDict

import numpy as np
import torch

from . import _C

# struct vspw_sgd_entry (include/vspw_hip.h)
_ENTRY = np.dtype([("p", "<u8"), ("g", "<u8"), ("buf", "<u8"), ("n", "<i8"), ("chunk0", "<i8"), ("lr", "<f4"),
                   ("wd", "<f4"), ("mult", "<i4"), ("first", "<i4")])


class SGD(torch.optim.Optimizer):
    def __init__(self, params, lr=0.02, momentum=0.0, weight_decay=0.0):
        defaults = dict(lr=lr, momentum=momentum, weight_decay=weight_decay)
        groups = []
        for g in params if isinstance(params, (list, tuple)) and params and isinstance(params[0], dict) else [
                {"params": params}]:
            g = dict(g)
            plist = list(g["params"])
            mult = OrderedDict()
            for p in plist:
                mult[p] = mult.get(p, 0) + 1
            g["params"] = list(mult.keys())
            g["mult"] = list(mult.values())
            groups.append(g)
        super().__init__(groups, defaults)

    @torch.no_grad()
    def step(self, closure=None):
        """One multi-tensor launch (vspw_sgd_multi) updates every parameter; per-parameter records (pointers, size,
        lr, weight decay, multiplicity) are rebuilt each step because autograd hands out fresh gradient tensors."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        by_mom = {}
        for group in self.param_groups:
            lr, wd, mom = float(group["lr"]), float(group["weight_decay"]), float(group["momentum"])
            for p, mult in zip(group["params"], group["mult"]):
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("vspw SGD runs on the GPU only (no CPU fallback)")
                g = p.grad
                if g.stride() != p.stride():
                    g = torch.empty_like(p).copy_(g)
                st = self.state[p]
                first = "momentum_buffer" not in st
                if first:
                    st["momentum_buffer"] = torch.empty_like(p)
                by_mom.setdefault((mom, p.device), []).append((p, g, st["momentum_buffer"], lr, wd, int(mult), first))
        chunk = int(_C.query("vspw_sgd_chunk_elems"))
        for (mom, dev), items in by_mom.items():
            rec = np.zeros(len(items), dtype=_ENTRY)
            c0 = 0
            for i, (p, g, buf, lr, wd, mult, first) in enumerate(items):
                n = p.numel()
                rec[i] = (p.data_ptr(), g.data_ptr(), buf.data_ptr(), n, c0, lr, wd, mult, 1 if first else 0)
                c0 += (n + chunk - 1) // chunk
            table = torch.from_numpy(rec.view(np.uint8)).to(dev, non_blocking=True)
            _C.call("vspw_sgd_multi", ctypes.c_void_p(table.data_ptr()), len(items), c0, mom,
                    ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            self._keepalive = (table, items)  # until the next step: the launch is asynchronous
        return loss


def create_optimizers(model, lr, weight_decay=1e-4, momentum=0.9, fix=False):
    """The four groups of train_clip2.py:215-236 (encoder at 0.1x lr; bias-named parameters without weight decay)."""
    if fix:
        groups = [{"params": model.get_10x_lr_params(), "lr": lr, "weight_decay": weight_decay},
                  {"params": model.get_10x_lr_params_bias(), "lr": lr, "weight_decay": 0}]
    else:
        groups = [{"params": model.get_1x_lr_params(), "lr": lr * 0.1, "weight_decay": weight_decay},
                  {"params": model.get_10x_lr_params(), "lr": lr, "weight_decay": weight_decay},
                  {"params": model.get_1x_lr_params_bias(), "lr": lr * 0.1, "weight_decay": 0},
                  {"params": model.get_10x_lr_params_bias(), "lr": lr, "weight_decay": 0}]
    return SGD(groups, lr=lr, momentum=momentum, weight_decay=weight_decay)


def adjust_learning_rate(optimizer, cur_iter, max_iters, lr, lr_pow=0.9, fix=False):
    """Poly schedule of train_clip2.py:239-252."""
    running = lr * ((1.0 - float(cur_iter) / max_iters) ** lr_pow)
    scales = [1.0, 1.0] if fix else [0.1, 1.0, 0.1, 1.0]
    for g, s in zip(optimizer.param_groups, scales):
        g["lr"] = running * s
    return running
