"""Diagnostic: two processes on one GPU hammer the peer exchange; print what a wrong total looks like."""
import os, sys, time
import numpy as np
import torch
import torch.multiprocessing as mp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def worker(rank, world, port, q, iters):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0", VSPW_DIST_TIMEOUT_S="90", VSPW_SHARED_GPU_TEST="1")
    from cvpr2021_vspw_implement_amd import distributed as vdist
    from cvpr2021_vspw_implement_amd.peer_exchange import PeerExchange
    vdist.init_from_env(backend="gloo")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    xc = PeerExchange(timeout_s=20.0)
    assert xc.ok, xc.why
    rs = np.random.RandomState(5)
    own = np.random.RandomState(100 + rank)
    log = []
    hist = {}
    for k in range(iters):
        n = int(rs.choice([1, 2, 3, 128, 512, 1024, 4096, 8192]))
        # value pattern: rank r contributes (k + 1) * (r + 1) + i * 1e-3: totals identify the exchange index
        i = torch.arange(n, dtype=torch.float64, device=dev) * 0.5  # (exactly representable: totals are exact)
        mine = (k + 1.0) * (rank + 1) + i
        t = mine.clone()
        if own.rand() < 0.02:
            time.sleep(0.003)
        if own.rand() < 0.05:
            torch.cuda.synchronize()
        xc.all_reduce(t)
        want = (k + 1.0) * (world * (world + 1) / 2.0) + world * i   # exact in fp64
        if not torch.equal(t, want):
            bad = (t != want).nonzero().flatten()
            seen = ((t[bad] - want[bad])).cpu().numpy()
            log.append((k, n, int(bad.numel()), int(bad.min()), int(bad.max()), np.unique(np.round(seen, 3))[:6].tolist()))
        hist[k] = n
    xc.check()
    q.put((rank, log))
    xc.close()
    torch.distributed.destroy_process_group()

if __name__ == "__main__":
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
    world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, world, port, q, iters)) for r in range(world)]
    [p.start() for p in ps]
    res = dict(q.get(timeout=300) for _ in range(world))
    [p.join(30) for p in ps]
    print("world", world, "iters", iters)
    for r in range(world):
        print("rank", r, "bad exchanges:", len(res[r]))
        for e in res[r][:12]:
            print("   k=%d n=%d bad=%d [%d..%d] total - expected: %s" % e)
