"""Host side of the peer statistics exchange (csrc/exchange.hip): the per-BatchNorm cross-rank sum of
SynchronizedBatchNorm (reference models/sync_batchnorm/batchnorm.py:110-131, comm.py:59-137) through hipIpc-mapped
arenas and ONE single-workgroup kernel per exchange instead of one RCCL all-reduce (224 per TCB-PSP step, each on the
critical path).

Start-up (collective over the process group, every rank takes the same decisions):
  1. each rank allocates its arena (uncached device memory) and exports an IPC handle;
  2. handles + host names are all-gathered; ranks on other hosts -> the exchange stays off (RCCL path);
  3. every rank maps its peers' arenas; "everybody could map everybody" is agreed with a MIN all-reduce;
  4. a self-test runs 48 exchanges of known patterns of varying length and checks the exact totals; agreed again.
Only if every step succeeded on every rank is the exchange used; otherwise the caller keeps using torch.distributed.
"""
import ctypes
import os
import socket
import sys

import torch
import torch.distributed as dist

from . import _C

_vp = ctypes.c_void_p


class PeerExchange:
    SLOT_DOUBLES = 8192  # 2 * C for C <= 4096 (the widest BatchNorm of the path has 2048 channels)

    def __init__(self, group=None, timeout_s=None):
        from . import distributed as vdist

        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.timeout_s = float(os.environ.get("VSPW_PEER_TIMEOUT_S", 20.0 if timeout_s is None else timeout_s))
        self.dev = torch.device("cuda", torch.cuda.current_device())
        self.ok = False
        self.why = ""
        self.own = None
        self.opened = []
        self.exchanges = 0
        self._closed = False
        lib = _C.load()
        nbytes = lib.vspw_xchg_arena_bytes(self.world, self.SLOT_DOUBLES)
        local_ok = nbytes > 0
        handle = ctypes.create_string_buffer(lib.vspw_xchg_handle_bytes())
        if local_ok:
            own = _vp()
            if lib.vspw_xchg_alloc(nbytes, ctypes.byref(own), handle) == 0:
                self.own = own.value
            else:
                local_ok = False
                self.why = "arena allocation / IPC export failed (hipError %d)" % lib.vspw_last_hip_error()
        else:
            self.why = "world size %d not supported by the exchange" % self.world
        # 2. gather handles
        info = (bytes(handle.raw), socket.gethostname(), bool(local_ok))
        infos = [info]
        if self.world > 1:
            infos = [None] * self.world
            dist.all_gather_object(infos, info, group=group)
        same_host = all(i[1] == infos[0][1] for i in infos)
        if not same_host:
            self.why = "ranks span several hosts"
        can = local_ok and same_host and all(i[2] for i in infos)
        # 3. map the peers
        self.arenas = (_vp * self.world)()
        if can:
            for r, (h, _, _) in enumerate(infos):
                if r == self.rank:
                    self.arenas[r] = self.own
                    continue
                p = _vp()
                if lib.vspw_xchg_open(h, ctypes.byref(p)) != 0:
                    can = False
                    self.why = "hipIpcOpenMemHandle of rank %d's arena failed (hipError %d)" % (r, lib.vspw_last_hip_error())
                    break
                self.opened.append(p.value)
                self.arenas[r] = p.value
        can = self._agree(can, vdist)
        if can:
            self.counter = torch.zeros(1, dtype=torch.int64, device=self.dev)
            self.status = torch.zeros(1, dtype=torch.int32, device=self.dev)
            torch.cuda.synchronize()
            if self.world > 1:
                dist.barrier(group=group)  # every arena is zeroed and mapped before the first flag is written
            can = self._agree(self._self_test(), vdist)
            if not can and not self.why:
                self.why = "self-test failed"
        self.ok = bool(can)
        if not self.ok:
            if self.rank == 0:
                sys.stderr.write("peer statistics exchange disabled (%s): SyncBN statistics go through "
                                 "torch.distributed\n" % (self.why or "a peer could not set it up"))
            self.close()

    def _agree(self, flag, vdist):
        if self.world == 1:
            return bool(flag)
        t = torch.tensor([1.0 if flag else 0.0], device=self.dev)
        vdist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(t.item() > 0.5)

    def _launch(self, t, timeout_s):
        _C.call("vspw_xchg_allreduce_f64", _vp(t.data_ptr()), t.numel(), ctypes.cast(self.arenas, _vp), self.world,
                self.rank, _vp(self.counter.data_ptr()), self.SLOT_DOUBLES, ctypes.c_double(timeout_s),
                _vp(self.status.data_ptr()), _vp(torch.cuda.current_stream().cuda_stream))

    def _self_test(self):
        """48 exchanges of rank-dependent patterns (lengths 1 ... SLOT_DOUBLES): totals must be exact on every rank."""
        try:
            w = self.world
            # test hook (tests/test_bench_gpu.py): this rank reports a failed self-test - EVERY rank must then fall back
            # failure injection for tests/test_bench_gpu.py: honoured in the shared-GPU TEST MODE only
            if (os.environ.get("VSPW_PEER_SELFTEST_FAIL_RANK") == str(self.rank)
                    and (os.environ.get("VSPW_SHARED_GPU_TEST") == "1" or os.environ.get("VSPW_BENCH_SHARED_GPU") == "1")):
                self.why = "self-test: failure injected on rank %d (VSPW_PEER_SELFTEST_FAIL_RANK)" % self.rank
            for k in range(48):
                n = [1, 2, 130, 1024, 4096, self.SLOT_DOUBLES][k % 6]
                base = torch.arange(n, dtype=torch.float64, device=self.dev) * 0.5 + k
                mine = base * (self.rank + 1) + self.rank * 0.25
                self._launch(mine, 5.0)
                want = base * (w * (w + 1) / 2.0) + 0.25 * (w * (w - 1) / 2.0)
                if not torch.equal(mine, want):
                    self.why = "self-test: wrong total in round %d" % k
                    return False
            if int(self.status.item()) != 0:
                self.why = "self-test: a wait timed out"
                return False
            return not self.why.startswith("self-test: failure injected")  # (the rounds above still ran: peers wait in them)
        except Exception as e:  # noqa: BLE001
            self.why = "self-test raised %r" % (e,)
            return False

    def usable(self, t):
        return self.ok and t.dtype == torch.float64 and t.is_contiguous() and 0 < t.numel() <= self.SLOT_DOUBLES

    def all_reduce(self, t):
        """In-place sum over the ranks (stream-ordered, capturable; no host synchronisation)."""
        self._launch(t, self.timeout_s)
        self.exchanges += 1

    def bn_finalize(self, sums, c, count, gamma, beta, running_mean, running_var, momentum, eps, mean, invstd, scale,
                    shift, clamp_var):
        """all_reduce(sums [2][c]) + the BatchNorm finalisation that consumes the totals, one launch."""
        p = lambda t: None if t is None else _vp(t.data_ptr())  # noqa: E731
        _C.call("vspw_xchg_bn_finalize", _vp(sums.data_ptr()), int(c), ctypes.cast(self.arenas, _vp), self.world,
                self.rank, _vp(self.counter.data_ptr()), self.SLOT_DOUBLES, ctypes.c_double(self.timeout_s),
                _vp(self.status.data_ptr()), ctypes.c_double(count), p(gamma), p(beta), p(running_mean),
                p(running_var), float(momentum), float(eps), p(mean), p(invstd), p(scale), p(shift),
                1 if clamp_var else 0, _vp(torch.cuda.current_stream().cuda_stream))
        self.exchanges += 1

    def check(self):
        """Raise if any exchange since start-up gave up waiting for a peer (synchronises: call at a sync point)."""
        if self.ok and int(self.status.item()) != 0:
            raise RuntimeError("peer statistics exchange: a rank did not arrive within %.0f s (results are NaN)"
                               % self.timeout_s)

    def close(self):
        """Collective over the group (like the constructor): EVERY rank runs the barrier, whether or not its own arena
        allocation succeeded - a rank that skipped it would pair its next collective with the others' barrier."""
        if self._closed:
            return
        self._closed = True
        lib = _C.load()
        try:
            torch.cuda.synchronize()
        except Exception:  # noqa: BLE001
            pass
        for p in self.opened:
            lib.vspw_xchg_close(_vp(p))
        self.opened = []
        if self.world > 1 and dist.is_initialized():
            try:
                dist.barrier(group=self.group)  # peers have unmapped before the memory goes away
            except Exception:  # noqa: BLE001
                pass
        if self.own is not None:
            lib.vspw_xchg_free(_vp(self.own))
            self.own = None
        self.ok = False
