"""SGD for the clip heads: torch.optim.SGD semantics, executed by one HIP kernel per parameter.

The reference builds four parameter groups from generators that yield a parameter once per enclosing module
(models/clip_psp.py:99-135), and torch.optim.SGD (a Python loop in the pinned 1.3.1) then applies the momentum
update once per occurrence, with the weight decay added to p.grad IN PLACE each time.  This optimizer keeps that
behaviour — group['params'] may contain duplicates — but folds the k occurrences of a parameter into ONE kernel launch
that applies the update k times in registers.
"""
import ctypes
from collections import OrderedDict

import numpy as np
import torch

from . import _C

# struct vspw_sgd_entry (include/vspw_hip.h)
_ENTRY = np.dtype([("p", "<u8"), ("g", "<u8"), ("buf", "<u8"), ("n", "<i8"), ("chunk0", "<i8"), ("lr", "<f4"),
                   ("wd", "<f4"), ("mult", "<i4"), ("first", "<i4"), ("lr_slot", "<i4"), ("reserved", "<i4")])


def _same_element_order(a, b):
    """Do two dense tensors of equal shape store their elements in the same order?  Strides of size-1 dimensions are
    arbitrary (a [K,C,1,1] weight is both contiguous and channels_last), so only the others are compared."""
    return a.shape == b.shape and all(sa == sb for n, sa, sb in zip(a.shape, a.stride(), b.stride()) if n != 1)


class SGD(torch.optim.Optimizer):
    """torch.optim.SGD(momentum, weight_decay) of the pinned PyTorch 1.3.1 (README.md:13), duplicates included: a
    parameter listed k times in a group is updated k times per step, the weight-decay term accumulating in the
    gradient across the k applications (1.3.1 adds it in place: d_p.add_(weight_decay, p.data)).

    In memory group['params'] is de-duplicated (multiplicities in group['mult'], the original sequence in
    group['order']); state_dict() / load_state_dict() speak torch.optim.Optimizer's own layout with the duplicates
    expanded, i.e. `opt_epoch_N.pth` files are interchangeable with the reference's (train_clip2.py:179-189,347-357):
    a run can be resumed from the reference's optimizer checkpoint and vice versa."""

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
            pos = {id(p): i for i, p in enumerate(g["params"])}
            g["order"] = [pos[id(p)] for p in plist]  # the generator's sequence, duplicates included
            groups.append(g)
        super().__init__(groups, defaults)
        self._buckets = {}  # (momentum, device) -> {"key", "table", "pinned"}: one parameter table per launch
        self._lr_dev = None
        self._lr_host = None
        self._graph_keepalive = []

    _PRIVATE = ("params", "mult", "order")

    def state_dict(self):
        """torch.optim.Optimizer.state_dict() of an SGD built on the ORIGINAL parameter lists (duplicates included):
        parameters are numbered in listing order; a parameter repeated inside a group carries the index of its LAST
        occurrence there (torch's dict-comprehension packing), one already seen in an earlier group keeps that group's
        index; the counter advances over duplicates too; state is keyed by the packed index."""
        mapping, start, groups = {}, 0, []
        for g in self.param_groups:
            expanded = [g["params"][i] for i in g["order"]]
            mapping.update({id(p): i for i, p in enumerate(expanded, start) if id(p) not in mapping})
            packed = {k: v for k, v in g.items() if k not in self._PRIVATE}
            packed["params"] = [mapping[id(p)] for p in expanded]
            start += len(expanded)
            groups.append(packed)
        state = {mapping[id(p)]: v for p, v in self.state.items() if id(p) in mapping}
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, state_dict):
        """Accepts the layout above - what the reference's torch.optim.SGD wrote - and this class's round-2 layout
        (de-duplicated lists with 'mult')."""
        saved = state_dict["param_groups"]
        if len(saved) != len(self.param_groups):
            raise ValueError("loaded state dict has a different number of parameter groups")
        if all("mult" in g for g in saved):
            # torch rebuilds every param_group from the saved dict (keeping only this optimizer's 'params'): the private
            # keys computed in __init__ - 'order', which the round-2 layout never stored, and 'mult' - are put back
            keep = [(g["mult"], g["order"]) for g in self.param_groups]
            for (mult, _), sg in zip(keep, saved):
                if list(sg["mult"]) != list(mult):
                    raise ValueError("loaded state dict lists different parameter multiplicities than this optimizer")
            super().load_state_dict(state_dict)
            for g, (mult, order) in zip(self.param_groups, keep):
                g["mult"], g["order"] = mult, order
            self._buckets = {}
            return
        by_index = {}
        for g, sg in zip(self.param_groups, saved):
            expanded = [g["params"][i] for i in g["order"]]
            if len(sg["params"]) != len(expanded):
                raise ValueError("loaded state dict contains a parameter group that doesn't match the size of "
                                 "optimizer's group")
            for idx, p in zip(sg["params"], expanded):
                if by_index.setdefault(idx, p) is not p:
                    raise ValueError("loaded state dict maps index %d to two different parameters" % idx)
            g.update({k: v for k, v in sg.items() if k not in self._PRIVATE})
        self.state.clear()
        for idx, st in state_dict["state"].items():
            p = by_index[int(idx)]
            new = {}
            for k, v in st.items():
                if torch.is_tensor(v) and v.shape == p.shape:
                    # the fused kernel walks parameter, gradient and momentum in the parameter's element order
                    v = torch.empty_like(p).copy_(v.to(device=p.device, dtype=p.dtype))
                new[k] = v
            self.state[p] = new
        self._buckets = {}

    def _collect(self):
        """[(momentum, device) -> [(p, g, buf, lr_slot, wd, mult)]]; momentum buffers are created zero-filled, which
        makes `buf = momentum*0 + g` on the first step exactly torch's `buf = clone(g)`."""
        by_mom = {}
        for gi, group in enumerate(self.param_groups):
            wd, mom = float(group["weight_decay"]), float(group["momentum"])
            for p, mult in zip(group["params"], group["mult"]):
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("vspw SGD runs on the GPU only (no CPU fallback)")
                g = p.grad
                if not _same_element_order(p, g):
                    g = torch.empty_like(p).copy_(g)
                st = self.state[p]
                if "momentum_buffer" not in st:
                    st["momentum_buffer"] = torch.zeros_like(p)
                by_mom.setdefault((mom, p.device), []).append((p, g, st["momentum_buffer"], gi, wd, int(mult)))
        return by_mom

    @torch.no_grad()
    def step(self, closure=None):
        """One multi-tensor launch (vspw_sgd_multi) updates every parameter.  The per-parameter records (pointers,
        size, weight decay, multiplicity, learning-rate slot) are kept on the device and only re-uploaded when a
        pointer changed (autograd hands out fresh gradient tensors in eager mode; under a captured hipGraph they are
        constant); learning rates travel separately through a [n_groups] device array."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        from . import ops

        ops.join_side_streams()  # weight gradients issued on the side stream
        by_mom = self._collect()
        if not by_mom:
            return loss
        chunk = int(_C.query("vspw_sgd_chunk_elems"))
        capturing = torch.cuda.is_current_stream_capturing()
        lrs = np.asarray([float(g["lr"]) for g in self.param_groups], dtype=np.float32)
        keep = []
        for (mom, dev), items in by_mom.items():
            if self._lr_dev is None or self._lr_dev.device != dev or self._lr_dev.numel() != lrs.size:
                self._lr_dev = torch.zeros(lrs.size, device=dev, dtype=torch.float32)
                self._lr_host = None
            if not capturing and (self._lr_host is None or not np.array_equal(self._lr_host, lrs)):
                # pageable -> device copy: the host buffer is consumed before the call returns (no race with the next
                # step's schedule update); under capture the caller refreshes the array with set_lrs() before replay
                self._lr_dev.copy_(torch.from_numpy(lrs.copy()))
                self._lr_host = lrs.copy()
            rec = np.zeros(len(items), dtype=_ENTRY)
            c0 = 0
            for i, (p, g, buf, slot, wd, mult) in enumerate(items):
                n = p.numel()
                rec[i] = (p.data_ptr(), g.data_ptr(), buf.data_ptr(), n, c0, 0.0, wd, mult, 0, slot, 0)
                c0 += (n + chunk - 1) // chunk
            bk = self._buckets.setdefault((mom, dev), {"key": None, "table": None, "pinned": None})
            key = rec.tobytes()
            if bk["key"] != key:
                host = torch.from_numpy(rec.view(np.uint8).copy())
                if capturing:
                    # a memcpy node needs a source that outlives the graph: a pinned staging buffer that was allocated
                    # by an earlier eager step (hipHostMalloc is not allowed while a stream is capturing)
                    if bk["pinned"] is None or bk["pinned"].numel() != host.numel():
                        raise RuntimeError("vspw SGD: run at least one eager step before capturing a hipGraph")
                    bk["pinned"].copy_(host)
                    bk["table"] = torch.empty(host.numel(), device=dev, dtype=torch.uint8)
                    bk["table"].copy_(bk["pinned"], non_blocking=True)
                    self._graph_keepalive.append((bk["pinned"], bk["table"]))
                    bk["pinned"] = None  # owned by the graph from now on; a later capture stages through a new one
                else:
                    if bk["pinned"] is None or bk["pinned"].numel() != host.numel():
                        bk["pinned"] = torch.empty(host.numel(), dtype=torch.uint8).pin_memory()
                    bk["table"] = host.to(dev)
                bk["key"] = key
            _C.call("vspw_sgd_multi", ctypes.c_void_p(bk["table"].data_ptr()), len(items), c0, mom,
                    ctypes.c_void_p(self._lr_dev.data_ptr()),
                    ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            keep.append(items)
        self._keepalive = keep  # until the next step: the launches are asynchronous
        ops.invalidate_inference_cache()  # folded conv+BN weights derive from the parameters just rewritten
        return loss

    def set_lrs(self):
        """Upload the groups' current learning rates (call before replaying a captured step)."""
        if self._lr_dev is not None:
            lrs = np.asarray([float(g["lr"]) for g in self.param_groups], dtype=np.float32)
            if self._lr_host is None or not np.array_equal(self._lr_host, lrs):
                self._lr_dev.copy_(torch.from_numpy(lrs.copy()))
                self._lr_host = lrs.copy()


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
