"""N>1 path on real GPU tensors: two ranks (gloo backend, both on cuda:0 — the GPU box has one device, and RCCL refuses
two ranks on one device) run the HIP kernels with SyncBN + the bucketed gradient reducer on half a batch each; the
result must equal one process on the full batch (SynchronizedBatchNorm / DataParallel semantics of the reference)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = [pytest.mark.gpu, pytest.mark.plumbing]  # plumbing: collected last (tests/conftest.py)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(dev):
    from helpers import build, load_det, zero_dropout

    # bottleneck encoder: exercises the fused BN-backward front end (ops.BNLink) and the folded skip gradient too
    mod = build("seg", "resnet50dilated", "ppm_deepsup", 2048)
    load_det(mod)
    zero_dropout(mod)
    return mod.to(dev).train()


def _data():
    from oracle.det_init import det_input, det_labels

    img = det_input("sync:img", (4, 3, 65, 65), seed=11)
    lab = det_labels("sync:lab", (4, 1, 65, 65), 124, seed=11, ignore_frac=0.0)
    return img, lab


def _worker(rank, world, port, q, backend="gloo", peer="1"):
    local = rank if backend == "nccl" else 0
    os.environ["VSPW_SYNCBN_PEER"] = peer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(local), HSA_ENABLE_IPC_MODE_LEGACY="0", VSPW_DIST_TIMEOUT_S="90")
    if backend == "gloo":
        os.environ["VSPW_SHARED_GPU_TEST"] = "1"  # host-staged collectives (distributed.all_reduce)
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    sys.path.insert(0, os.path.dirname(here))
    from cvpr2021_vspw_implement_amd import distributed as vdist
    from cvpr2021_vspw_implement_amd import watchdog

    wd = watchdog.make(True, 100.0)
    wd.phase("init process group")
    vdist.init_from_env(backend=backend)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(local)
    wd.phase("build + broadcast")
    mod = _build(dev)
    wrapped = vdist.DataParallelOverRCCL(mod, bucket_mb=4.0, sync_bn=True)
    img, lab = _data()
    wd.phase("forward + backward + gradient averaging")
    sl = slice(rank * 2, rank * 2 + 2)
    loss, acc = wrapped({"img_data": torch.from_numpy(img[sl]).to(dev), "seg_label": torch.from_numpy(lab[sl]).to(dev)})
    loss.backward()
    wrapped.finish_gradients()
    torch.cuda.synchronize()
    out = {"loss": loss.item(),
           "grads": {k: p.grad.double().norm().item() for k, p in mod.named_parameters() if p.grad is not None},
           "peer": wrapped.exchange is not None, "exchanges": wrapped.exchange.exchanges if wrapped.exchange else 0,
           "grad_sample": mod.encoder.layer1[0].conv1.weight.grad.detach().cpu().numpy().copy(),
           "rm": mod.encoder.layer3[0].bn1.running_mean.cpu().numpy(),
           "rv": mod.encoder.layer3[0].bn1.running_var.cpu().numpy()}
    wd.phase("report + destroy process group")
    wrapped.check_exchange()
    q.put((rank, out))
    if wrapped.exchange is not None:
        wrapped.exchange.close()
    torch.distributed.destroy_process_group()
    wd.phase("exit")
    wd.stop()


def test_two_ranks_equal_one_full_batch(dev):
    """SyncBN statistics through the hipIpc PEER EXCHANGE (csrc/exchange.hip; both processes map each other's arena on
    the shared device), then through torch.distributed: both equal one process on the full batch, and each other bit for
    bit (two contributions: a + b is the same number in either order)."""
    a = _two_ranks(dev, "gloo", peer="1")
    assert a[0]["peer"] and a[1]["peer"] and a[0]["exchanges"] > 100, (a[0]["peer"], a[0]["exchanges"])
    b = _two_ranks(dev, "gloo", peer="0")
    assert not b[0]["peer"] and b[0]["exchanges"] == 0
    for r in (0, 1):
        assert a[r]["loss"] == b[r]["loss"]
        assert np.array_equal(a[r]["grad_sample"], b[r]["grad_sample"])
        assert np.array_equal(a[r]["rv"], b[r]["rv"])


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one device per rank (single-GPU box)")
def test_two_ranks_rccl_equal_one_full_batch(dev):
    """Same check over RCCL / xGMI with one process per GPU - runs wherever two devices are visible."""
    _two_ranks(dev, "nccl")


def _two_ranks(dev, backend, peer="1"):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, backend, peer)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=150) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # single process, full batch (no labels ignored, so mean-of-rank-means == full-batch mean)
    from cvpr2021_vspw_implement_amd import ops

    ops.set_sync_bn(False)
    mod = _build(dev)
    img, lab = _data()
    loss, acc = mod({"img_data": torch.from_numpy(img).to(dev), "seg_label": torch.from_numpy(lab).to(dev)})
    loss.backward()
    ref = {k: p.grad.double().norm().item() for k, p in mod.named_parameters() if p.grad is not None}
    mean_loss = 0.5 * (res[0]["loss"] + res[1]["loss"])
    assert abs(mean_loss - loss.item()) < 2e-5 * abs(loss.item()), (mean_loss, loss.item())
    scale = max(ref.values())
    for r in (0, 1):
        worst = max(abs(res[r]["grads"][k] - v) / max(v, 1e-3 * scale) for k, v in ref.items())
        assert worst < 3e-2, (r, worst)
        assert np.abs(res[r]["rm"] - mod.encoder.layer3[0].bn1.running_mean.cpu().numpy()).max() < 1e-5
        assert np.abs(res[r]["rv"] - mod.encoder.layer3[0].bn1.running_var.cpu().numpy()).max() < 1e-5
    # both ranks hold identical (averaged) gradients
    assert max(abs(res[0]["grads"][k] - res[1]["grads"][k]) / max(v, 1e-3 * scale) for k, v in ref.items()) < 1e-6
    return res


def _protocol_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0", VSPW_DIST_TIMEOUT_S="90", VSPW_SHARED_GPU_TEST="1")
    import sys
    import time

    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    from cvpr2021_vspw_implement_amd import distributed as vdist
    from cvpr2021_vspw_implement_amd import watchdog
    from cvpr2021_vspw_implement_amd.peer_exchange import PeerExchange

    wd = watchdog.make(True, 100.0)
    wd.phase("init")
    vdist.init_from_env(backend="gloo")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    xc = PeerExchange(timeout_s=20.0)
    assert xc.ok, xc.why
    wd.phase("stress: 1500 exchanges, ragged sizes, random stalls")
    rs = np.random.RandomState(5)          # same sequence of sizes on both ranks
    own = np.random.RandomState(100 + rank)  # rank-private stalls
    bad = 0
    side = torch.cuda.Stream()
    main = torch.cuda.current_stream()
    for k in range(1500):
        n = int(rs.choice([1, 2, 3, 128, 512, 1024, 4096, 8192]))
        vals = [torch.from_numpy(np.random.RandomState(1000 * r + k).randn(n)).to(dev) for r in range(world)]
        t = vals[rank].clone()
        if own.rand() < 0.02:
            time.sleep(0.003)              # host stall: the peer's kernel waits for this rank's flag
        if own.rand() < 0.05:
            torch.cuda.synchronize()
        if k % 7 == 3:  # every seventh exchange from a second stream, ordered after / before the main stream's work
            side.wait_stream(main)
            with torch.cuda.stream(side):
                xc.all_reduce(t)
            main.wait_stream(side)
        else:
            xc.all_reduce(t)
        want = vals[0].clone()
        for r in range(1, world):
            want += vals[r]
        bad += int(not torch.equal(t, want))
    xc.check()
    wd.phase("dead peers: only rank 0 launches, it must give up after 1 s with NaN + status")
    gave_up = None
    vdist.dist.barrier()
    if rank == 0:
        xc.timeout_s = 1.0
        t = torch.ones(64, dtype=torch.float64, device=dev)
        t0 = time.time()
        xc.all_reduce(t)
        torch.cuda.synchronize()
        took = time.time() - t0
        try:
            xc.check()
            gave_up = False
        except RuntimeError:
            gave_up = bool(torch.isnan(t).all().item()) and 0.9 < took < 10.0
    vdist.dist.barrier()
    wd.phase("report")
    q.put((rank, {"bad": bad, "gave_up": gave_up, "exchanges": xc.exchanges}))
    xc.close()
    torch.distributed.destroy_process_group()
    wd.stop()


@pytest.mark.parametrize("world", [2, 4])
def test_peer_exchange_protocol_between_processes(dev, world):
    """csrc/exchange.hip between 2 and 4 PROCESSES that map each other's arenas (hipIpc) on the one device of the box: 1 500
    exchanges of ragged lengths with host stalls, stream switches and device syncs thrown in give the exact fp64 totals
    on both ranks; a peer that never arrives costs the waiting rank its timeout - not the GPU."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_protocol_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert all(res[r]["bad"] == 0 for r in range(world)), res
    assert res[0]["gave_up"] is True
    assert res[0]["exchanges"] == 1501 and all(res[r]["exchanges"] == 1500 for r in range(1, world))
