"""One process per GPU over torch.distributed (backend "nccl" = RCCL on ROCm, xGMI between the 8 MI355X of a node).

Replaces the reference's single-process nn.DataParallel + thread-based SyncBN (train_clip2.py:359-364,
models/sync_batchnorm/comm.py): clips shard across ranks with no data-path exchange; the only collectives are
  * the gradient all-reduce (282 MB fp32 for TCB-PSP R101), issued per ~25 MB bucket from post-accumulate-grad hooks
    so that it overlaps the rest of the backward pass (xGMI is point-to-point; ring all-reduce is per-link bound,
    so few large buckets beat many small ones);
  * the per-layer BatchNorm statistics all-reduce (2C doubles forward, 2C backward) that gives
    SynchronizedBatchNorm semantics (ops.set_sync_bn).
DataParallel averages the per-replica losses (train_clip2.py:98), so gradients are averaged, not summed.
"""
import math
import os

import torch
import torch.distributed as dist

from . import ops


def shared_gpu_test():
    """TEST MODE (tests/test_bench_gpu.py, tests/test_drivers_gpu.py): VSPW_SHARED_GPU_TEST=1 (or bench.py's
    VSPW_BENCH_SHARED_GPU=1) lets several ranks share the visible device(s) over the gloo backend - RCCL refuses two
    ranks on one GPU - so that the multi-rank host path can run end to end on a 1-GPU box.  Never a measurement."""
    return os.environ.get("VSPW_SHARED_GPU_TEST") == "1" or os.environ.get("VSPW_BENCH_SHARED_GPU") == "1"


class _Completed:
    """Handle of a collective that has already finished (test mode below)."""

    def wait(self):
        return True


def all_reduce(t, op=None, group=None, async_op=False):
    """torch.distributed.all_reduce.  Production (RCCL): passed straight through.  In the shared-GPU TEST MODE a device
    tensor is staged through host memory by this function - .cpu() (ordered after the tensor's producers on the current
    stream), a gloo all-reduce of the host copy, a copy back - instead of handing the device tensor to
    ProcessGroupGloo, whose own side streams / pinned staging pool / worker threads are one more moving part between
    two processes that share a device and are not what that mode is there to exercise (launcher, rendezvous,
    broadcast, SyncBN exchange points, bucketing, max-over-ranks timing, rank-0 reporting)."""
    op = dist.ReduceOp.SUM if op is None else op
    if t.is_cuda and shared_gpu_test():
        host = t.detach().cpu()
        dist.all_reduce(host, op=op, group=group)
        t.copy_(host)
        return _Completed() if async_op else None
    return dist.all_reduce(t, op=op, group=group, async_op=async_op)


def broadcast(t, src=0, group=None):
    """torch.distributed.broadcast; host-staged in the shared-GPU test mode (see all_reduce)."""
    if t.is_cuda and shared_gpu_test():
        host = t.detach().cpu()
        dist.broadcast(host, src=src, group=group)
        t.copy_(host)
        return None
    return dist.broadcast(t, src=src, group=group)


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if shared_gpu_test() and torch.cuda.is_available():
        backend = backend or "gloo"
        local_rank = local_rank % torch.cuda.device_count()
    if (world > 1 or os.environ.get("VSPW_FORCE_COLLECTIVES") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if os.environ.get("VSPW_DIST_TIMEOUT_S"):  # collectives raise instead of blocking forever (test modes)
            import datetime

            kw["timeout"] = datetime.timedelta(seconds=float(os.environ["VSPW_DIST_TIMEOUT_S"]))
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank), **kw)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, local_rank, world


class GradReducer:
    """Bucketed, backward-overlapped gradient averaging for any nn.Module (device-agnostic host logic)."""

    def __init__(self, module, bucket_mb=25.0, group=None, force=False, find_unused_parameters=False):
        """find_unused_parameters=False (default, the contract of torch DDP's default too): every rank produces
        gradients for the same set of parameters each step - true for the VSPW heads, whose conditional paths
        (psp_weight, use_memory, clipocr_all) are configuration, not data, dependent.  True: a per-parameter
        "produced a gradient" flag is reduced with each bucket and read back (one device sync per bucket per step),
        so a parameter that got a gradient on ANY rank is updated on EVERY rank and one that got none anywhere keeps
        grad=None."""
        self.group = group
        self.timer = None  # bench.py diagnostics: dict(buckets=[(launch event, done event)], wait=[(e0, e1)]) or None
        self.find_unused = bool(find_unused_parameters)
        self._flag_cache = {}
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())
        params = [p for p in module.parameters() if p.requires_grad]
        seen, uniq = set(), []
        for p in params:
            if id(p) not in seen:
                seen.add(id(p))
                uniq.append(p)
        uniq.reverse()  # backward produces gradients roughly in reverse registration order
        cap = int(bucket_mb * 1024 * 1024 / 4)
        self.buckets = []
        cur, cur_n = [], 0
        for p in uniq:
            if cur and cur_n + p.numel() > cap:
                self.buckets.append(cur)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += p.numel()
        if cur:
            self.buckets.append(cur)
        self.flat, self.slots, self.owner = [], {}, {}
        self.nelem = []
        for bi, bucket in enumerate(self.buckets):
            n = sum(p.numel() for p in bucket)
            self.nelem.append(n)
            # the tail of the buffer carries one "this rank produced a gradient" flag per parameter, reduced with the
            # gradients: a parameter that got a gradient on ANY rank is updated on EVERY rank (replicas stay equal),
            # one that got none anywhere keeps grad=None (and is skipped by the optimizer, as in a single process)
            flat = torch.zeros(n + len(bucket), device=bucket[0].device, dtype=bucket[0].dtype)
            off = 0
            for p in bucket:
                # a view of the flat buffer with the parameter's own (dense, possibly channels_last) strides
                self.slots[id(p)] = flat[off:off + p.numel()].as_strided(p.shape, p.stride())
                self.owner[id(p)] = bi
                off += p.numel()
            self.flat.append(flat)
        self.pending = [len(b) for b in self.buckets]
        self.handles = [None] * len(self.buckets)
        self.hooks = []
        if self.active:
            for p in uniq:
                self.hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def broadcast_parameters(self, module, src=0):
        """Rank `src`'s parameters and buffers to every rank (the reference's DataParallel re-broadcasts the replicas
        every step, train_clip2.py:359-364; one process per GPU needs it once).  ONE broadcast per dtype over a
        flattened copy instead of ~1 300 tiny ones: a start-up of milliseconds over RCCL, and no per-message latency
        (40 ms delayed-ACK stalls were measured per small gloo message in the shared-GPU test mode: a minute in total)."""
        if not self.active:
            return
        by_dtype = {}
        seen = set()
        for t in list(module.parameters()) + list(module.buffers()):
            if (t.is_floating_point() or t.dtype == torch.long) and id(t) not in seen:
                seen.add(id(t))
                by_dtype.setdefault((t.dtype, t.device), []).append(t.data)
        for ts in by_dtype.values():
            flat = torch.cat([t.reshape(-1) for t in ts])  # reshape(-1): a copy in logical order for channels_last weights
            broadcast(flat, src=src, group=self.group)
            off = 0
            for t in ts:
                n = t.numel()
                t.copy_(flat[off:off + n].view(t.shape))
                off += n

    def _launch(self, bi):
        """Gather the bucket's gradients into its flat buffer with ONE multi-tensor copy, then start the all-reduce."""
        bucket = self.buckets[bi]
        ops.join_side_streams()  # weight gradients may still be in flight on the side stream
        have = [p for p in bucket if p.grad is not None and p.grad.data_ptr() != self.slots[id(p)].data_ptr()]
        if have:
            torch._foreach_copy_([self.slots[id(p)] for p in have], [p.grad for p in have])
        n = self.nelem[bi]
        pattern = tuple(p.grad is not None for p in bucket)
        for p, present in zip(bucket, pattern):
            if not present:
                self.slots[id(p)].zero_()
        if self.find_unused:
            cached = self._flag_cache.get(bi)
            if cached is None or cached[0] != pattern:  # pageable upload only when the pattern changes
                cached = (pattern, torch.tensor([1.0 if f else 0.0 for f in pattern]).to(self.flat[bi].device))
                self._flag_cache[bi] = cached
            self.flat[bi][n:].copy_(cached[1])
        if self.timer is not None:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
            self.timer.setdefault("launch", {})[bi] = e0
        self.handles[bi] = all_reduce(self.flat[bi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _on_grad(self, p):
        bi = self.owner[id(p)]
        self.pending[bi] -= 1
        if self.pending[bi] == 0:
            self._launch(bi)

    def wait(self):
        """Finish the outstanding all-reduces and write the averaged gradients back. Call before optimizer.step()."""
        if not self.active:
            return
        inv = 1.0 / self.world
        if self.timer is not None:
            w0 = torch.cuda.Event(enable_timing=True)
            w0.record()
        for bi, bucket in enumerate(self.buckets):
            if self.pending[bi] != 0:  # parameters that got no gradient this step: reduce what is there
                self._launch(bi)
            self.handles[bi].wait()
            if self.timer is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                self.timer.setdefault("buckets", []).append((self.timer["launch"].pop(bi), e1))
            self.flat[bi].mul_(inv)
            flags = None
            for j, p in enumerate(bucket):
                if p.grad is None:
                    if not self.find_unused:
                        continue  # contract: no rank produced one either
                    # no local gradient: take the average anyway if some other rank produced one (reads the reduced
                    # flags - a device sync, paid only when a local gradient is missing)
                    if flags is None:
                        flags = self.flat[bi][self.nelem[bi]:].tolist()
                    if flags[j] <= 0.0:
                        continue
                # the averaged gradient IS the bucket slot (same shape and strides as the parameter): no copy back.
                # The drivers call zero_grad() every step (train_clip2.py:87), which drops these views; without it
                # autograd accumulates into the slot in place and the next reduction averages (previous average +
                # new local gradient), which is still the average of the accumulated gradients.
                p.grad = self.slots[id(p)]
            self.pending[bi] = len(bucket)
            self.handles[bi] = None
        if self.timer is not None:
            w1 = torch.cuda.Event(enable_timing=True)
            w1.record()
            self.timer.setdefault("wait", []).append((w0, w1))


def _world():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def all_agree(flag, device=None):
    """True on every rank iff `flag` is true on every rank (one tiny MIN all-reduce + a host read; every rank must
    call it).  The drivers use it before choosing between a captured hipGraph and the eager step: a rank that replays a
    graph while a peer runs the step launch by launch would issue a different sequence of collectives."""
    if _world() <= 1:
        return bool(flag)
    dev = device if device is not None else (torch.device("cuda", torch.cuda.current_device())
                                             if (torch.cuda.is_available() and dist.get_backend() != "gloo") else "cpu")
    t = torch.tensor([1 if flag else 0], dtype=torch.int64, device=dev)
    all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()) == 1)


class PeerAbort(RuntimeError):
    """Raised by step_guard on the ranks whose own step was fine when another rank aborted."""


def step_guard(module, loss_value):
    """Called by the drivers where they read the loss (a device sync they pay anyway): a peer statistics exchange that
    timed out has overwritten BatchNorm totals with NaN (exchange.hip) - raise instead of training on, and never let a
    non-finite loss reach the running statistics / a checkpoint silently.
    With more than one rank the decision is COLLECTIVE: the loss is per rank, so a rank that raised alone would leave its
    peers blocked in the next step's first collective (or in checkpoint_barrier) until the process-group timeout.  Every
    rank contributes a "my step is bad" flag to one MAX all-reduce here; the rank(s) with the local problem raise their
    own error, every other rank raises PeerAbort - all of them in the same step."""
    err = None
    try:
        if hasattr(module, "check_exchange"):
            module.check_exchange()
    except (RuntimeError, ValueError) as e:
        err = e
    if err is None and not math.isfinite(loss_value):
        err = FloatingPointError("non-finite training loss (%r): aborting before it is written into a checkpoint" % (loss_value,))
    if _world() > 1 and getattr(module, "reducer", None) is not None and module.reducer.active:
        dev = next(module.parameters()).device
        t = torch.tensor([0 if err is None else 1], dtype=torch.int64, device=dev)
        all_reduce(t, op=dist.ReduceOp.MAX)
        if err is None and int(t.item()) != 0:
            err = PeerAbort("another rank aborted this training step (non-finite loss, uneven SyncBN batch or a timed-out "
                            "peer exchange there): stopping here too instead of waiting in the next collective")
    if err is not None:
        raise err


def checkpoint_barrier():
    """After the rank-0-only checkpoint: the other ranks wait HERE (process-group timeout) instead of spinning in the
    next step's peer exchange (VSPW_PEER_TIMEOUT_S, 20 s) while rank 0 is still inside torch.save."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


class DataParallelOverRCCL(torch.nn.Module):
    """Drop-in for `nn.DataParallel(module)` + `patch_replication_callback` in the clip drivers: same call
    signature (`module(feed_dict)` -> (loss, acc)), one process per GPU underneath."""

    def __init__(self, module, bucket_mb=25.0, sync_bn=True, force_collectives=False, find_unused_parameters=False,
                 sync_bn_clamp_var=False):
        """sync_bn=False: every rank normalises with its own batch statistics (plain nn.BatchNorm under DDP).
        sync_bn_clamp_var: see ops.set_sync_bn (the reference's multi-device clamp(var, eps) formula)."""
        super().__init__()
        self.module = module
        self.reducer = GradReducer(module, bucket_mb, force=force_collectives,
                                   find_unused_parameters=find_unused_parameters)
        self.reducer.broadcast_parameters(module)
        # SyncBN statistics: hipIpc peer exchange (peer_exchange.py) when every rank of the group could set it up and
        # its self-test passed everywhere, torch.distributed all-reduces otherwise (VSPW_SYNCBN_PEER=0: always)
        self._sync_bn = bool(sync_bn)
        self._sig_cache, self._sig_pending = {}, None
        self._sig_work = self._sig_host = self._sig_stream = None
        self.exchange = None
        self.exchange_why = "not requested"  # why SyncBN statistics go through torch.distributed instead (bench.py reports it)
        if sync_bn and self.reducer.active and torch.cuda.is_available() and os.environ.get("VSPW_SYNCBN_PEER", "1") == "1":
            from .peer_exchange import PeerExchange

            xc = PeerExchange()
            self.exchange = xc if xc.ok else None
            self.exchange_why = "" if xc.ok else (xc.why or "a peer could not set it up")
        ops.set_sync_bn(sync_bn and self.reducer.active, force=force_collectives, clamp_var=sync_bn_clamp_var,
                        exchange=self.exchange)

    def forward(self, *a, **k):
        self._check_equal_batches(a, k)
        return self.module(*a, **k)

    # ---- SyncBN needs the same batch shape on every rank -----------------------------------------------------------
    # SyncBN totals are finalised with count = local rows x ranks (the fused exchange + finalise kernel carries sums
    # only), which is the reference's sum over the replicas' real sizes (models/sync_batchnorm/batchnorm.py:110-131) only
    # when every rank holds the same batch shape - what the drivers' drop_last guarantees.  A caller that feeds uneven
    # batches gets a ValueError ON EVERY RANK instead of silently biased statistics.  The comparison is symmetric and
    # unconditional: EVERY rank contributes a 62-bit hash of its shape signature to one MAX all-reduce of (h, -h) on
    # EVERY training-mode forward (a rank whose batch changes mid-run - a short last batch - is caught although its
    # peers' signatures did not change; nobody ever enters a collective alone).  On a GPU the 16-byte all-reduce and its
    # copy to pinned host memory are stream-ordered and the result is read one call site later (finish_gradients - i.e.
    # before the optimizer can apply the biased step - the next forward, check_exchange), so the check never drains the
    # queue; on the host (gloo, tests) it is read at once.
    @staticmethod
    def _shape_signature(a, k):
        def shapes(o):
            if torch.is_tensor(o):
                return tuple(o.shape)
            if isinstance(o, dict):
                return tuple((kk, shapes(o[kk])) for kk in sorted(o, key=str) if torch.is_tensor(o[kk]) or isinstance(o[kk], (list, tuple, dict)))
            if isinstance(o, (list, tuple)):
                return tuple(shapes(v) for v in o)
            return None

        return (shapes(a), shapes(k))

    def _check_equal_batches(self, a, k):
        if not (self.training and self.reducer.active and self._sync_bn and dist.is_initialized() and dist.get_world_size() > 1):
            return
        self._verify_batch_check()  # the previous call's result, if nobody has read it yet
        import hashlib

        sig = self._shape_signature(a, k)
        src = self._sig_cache.get(sig)
        dev = next(self.module.parameters()).device
        if src is None:
            h = int.from_bytes(hashlib.blake2b(repr(sig).encode(), digest_size=8).digest(), "little") >> 2
            src = self._sig_cache[sig] = torch.tensor([h, -h], dtype=torch.int64).to(dev)
        if dev.type != "cuda" or shared_gpu_test():
            t = src.clone()
            all_reduce(t, op=dist.ReduceOp.MAX)
            self._sig_pending = (None, t.cpu(), sig)
            self._verify_batch_check()
            return
        if self._sig_work is None:
            self._sig_work = torch.empty(2, dtype=torch.int64, device=dev)
            self._sig_host = torch.empty(2, dtype=torch.int64).pin_memory()
            self._sig_stream = torch.cuda.Stream(device=dev)
        self._sig_work.copy_(src)
        all_reduce(self._sig_work, op=dist.ReduceOp.MAX)
        if torch.cuda.is_current_stream_capturing():
            return  # a captured step: the shapes are frozen with the graph (the capture's warm-up runs were checked)
        cur = torch.cuda.current_stream()
        self._sig_stream.wait_stream(cur)
        with torch.cuda.stream(self._sig_stream):
            self._sig_host.copy_(self._sig_work, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._sig_stream)
        cur.wait_stream(self._sig_stream)  # the work buffer is rewritten by the next call
        self._sig_pending = (ev, self._sig_host, sig)

    def _verify_batch_check(self):
        pend, self._sig_pending = self._sig_pending, None
        if pend is None:
            return
        ev, host, sig = pend
        if ev is not None:
            if torch.cuda.is_current_stream_capturing():  # no host waits inside a capture: keep it for the next call site
                self._sig_pending = pend
                return
            ev.synchronize()
        hi, neg_lo = (int(v) for v in host.tolist())
        if hi != -neg_lo:
            # every rank sees the same reduced pair, so every rank is here: a collective for the message is safe
            every = [None] * dist.get_world_size()
            try:
                dist.all_gather_object(every, sig)
            except Exception:  # noqa: BLE001 - the message is a courtesy, the error is not
                every = ["rank %d: %r" % (dist.get_rank(), sig)]
            raise ValueError("SyncBN over %d ranks needs the same batch shape on every rank (drop_last); got %r"
                             % (dist.get_world_size(), every))

    def finish_gradients(self):
        self._verify_batch_check()  # before the optimizer can apply a step taken on biased SyncBN statistics
        self.reducer.wait()

    def close(self):
        """Release the peer-exchange arenas / IPC mappings and take the exchange out of ops (collective: call on every
        rank, before destroy_process_group).  Building another wrapper afterwards starts from a clean state."""
        if self.exchange is not None:
            self.exchange.close()
            self.exchange = None
        ops.set_sync_bn(False)

    def check_exchange(self):
        """Raise if a peer statistics exchange timed out (device sync: call where the loss is read anyway)."""
        self._verify_batch_check()
        if self.exchange is not None:
            self.exchange.check()
