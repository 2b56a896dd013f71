"""bench.py's N > 1 path end to end on a 1-GPU box: `python bench.py --gpus 2` launches two ranks itself
(torch.distributed.run, 127.0.0.1 rendezvous); in the VSPW_BENCH_SHARED_GPU test mode both share the device over gloo
(RCCL refuses two ranks on one GPU), everything else is the production path: parameter broadcast, SyncBN statistics
exchange per BatchNorm, bucketed gradient averaging, barrier + max-over-ranks timing, one JSON line from rank 0."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.plumbing]  # plumbing: collected last (tests/conftest.py)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, env, timeout):
    """Run a bench command under a hard timeout; on expiry the child's whole process group is killed and what it wrote
    (bench.py's phase log / watchdog stacks on stderr) is returned for the assertion message."""
    import signal

    p = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                         start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
        return p.returncode, out, err
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        out, err = p.communicate()
        return -9, out, "TIMEOUT after %d s\n%s" % (timeout, err)


def test_bench_launches_two_ranks_and_reports_once(dev):
    # collectives raise after 90 s (gloo timeout), a rank without progress for 100 s dumps its stacks and exits 3
    # (cvpr2021_vspw_implement_amd/watchdog.py), the whole launch is killed after 150 s
    env = dict(os.environ, VSPW_BENCH_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", VSPW_DIST_TIMEOUT_S="90",
               VSPW_WATCHDOG_S="100")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--crop", "191", "--no-cpu-baseline", "--no-host-probe", "--no-kernel-timing"]
    rc, out, err = _run(cmd, env, 150)
    if rc != 0:  # one retry, with the evidence of the first attempt on record (a warning in the test report)
        import warnings

        warnings.warn("first attempt of the two-rank bench launch failed (rc %d):\n%s" % (rc, err[-4000:]))
        rc, out, err = _run(cmd, env, 150)
    assert rc == 0, err[-4000:]
    r = type("R", (), {"stdout": out, "stderr": err})()
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1
    assert out["scaling"] == "weak" and out["value"] > 0 and out["ms_per_step"] > 0
    cfg = out["config"]
    assert cfg["parallelism"] == "dp2" and cfg["global_batch_clips"] == 4 and cfg["sync_bn"] is True
    assert cfg["execution"] == "eager launches" and "gloo" in cfg["backend"]
    assert abs(out["value"] - 2 * 2 / (out["ms_per_step"] / 1e3)) < 0.05 * out["value"]  # whole-job clips/s
    assert 6.0 < out["last_loss"] < 8.0
    assert "191x191" in cfg["workload"] and "NOT the metric" in cfg["workload"] and out["e2e_mfma_frac"] is None
    assert "[rank 0" in r.stderr and "[rank 1" in r.stderr  # the phase log of both ranks


@pytest.mark.parametrize("method", ["clip_psp", "clip_ocr"])
def test_bench_eight_ranks_plumbing(dev, method):
    """The launch the driver's SCALE run will make - `bench.py --gpus 8` - with the eight ranks sharing this box's one
    GPU (test mode: gloo instead of RCCL, which refuses two ranks per device; `rccl_ranks` is therefore 0 here and 8 on
    an 8-GPU node) at 95x95 crops: parameter broadcast, 8-way peer statistics exchange per BatchNorm, bucketed
    gradient averaging, barrier + max-over-ranks timing, exactly ONE JSON line.  cfg 4 (TCB-OCR) is the configuration
    BASELINE.json names for 8 GPUs."""
    import time

    env = dict(os.environ, VSPW_BENCH_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", VSPW_DIST_TIMEOUT_S="90",
               VSPW_WATCHDOG_S="100")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--crop", "95",
           "--method", method, "--no-cpu-baseline", "--no-host-probe"]  # (kernel timing on: `collectives` needs it)
    t0 = time.time()
    rc, out, err = _run(cmd, env, 200)
    wall = time.time() - t0
    assert rc == 0, err[-4000:]
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    o = json.loads(lines[0])
    print("8 ranks on one GPU, %s at 95x95: %.0f s wall, %.1f ms/step, collectives %s" % (method, wall, o["ms_per_step"],
                                                                                         json.dumps(o["collectives"])))
    assert o["n_gpus"] == 8 and o["config"]["ranks"] == 8 and o["config"]["parallelism"] == "dp8"
    assert o["config"]["global_batch_clips"] == 16 and o["scaling"] == "weak"
    assert abs(o["value"] - 8 * 2 / (o["ms_per_step"] / 1e3)) < 0.05 * o["value"]
    c = o["collectives"]
    assert c["syncbn_exchanges_per_step"] > 200 and "peer exchange" in c["syncbn_exchange"] and c["grad_buckets"] >= 4
    assert 6.0 < o["last_loss"] < 8.0
    for r in range(8):
        assert "[rank %d" % r in err
    assert wall < 120.0, wall  # (a fresh box: ~20 s of that is eight concurrent `import torch`)


def test_peer_exchange_failure_on_one_rank_makes_every_rank_fall_back(dev):
    """Start-up of the hipIpc statistics exchange is collective: a self-test that fails on ONE rank
    (VSPW_PEER_SELFTEST_FAIL_RANK, a hook in peer_exchange._self_test) must switch EVERY rank to the torch.distributed
    path - a mixed group would pair an all-reduce with a peer kernel and hang.  The run completes and says so."""
    env = dict(os.environ, VSPW_BENCH_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", VSPW_DIST_TIMEOUT_S="90",
               VSPW_WATCHDOG_S="100", VSPW_PEER_SELFTEST_FAIL_RANK="1")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--crop", "95",
           "--no-cpu-baseline", "--no-host-probe"]
    rc, out, err = _run(cmd, env, 200)
    assert rc == 0, err[-4000:]
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    o = json.loads(lines[0])
    c = o["collectives"]
    assert c["syncbn_exchange"].startswith("torch.distributed all-reduce"), c["syncbn_exchange"]
    assert "peer statistics exchange disabled" in err
    assert c["syncbn_exchanges_per_step"] > 200 and 6.0 < o["last_loss"] < 8.0


def test_bench_refuses_more_ranks_than_devices(dev):
    import torch

    n = torch.cuda.device_count() + 1
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("VSPW_BENCH_SHARED_GPU", None)
    rc, out, err = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1"], env, 120)
    assert rc != 0 and "GPU(s) are visible" in err


def test_bench_reports_collective_diagnostics_with_a_one_rank_rccl_group(dev):
    """VSPW_FORCE_COLLECTIVES=1: the N > 1 code path (RCCL process group, SyncBN exchange per BatchNorm, bucketed
    all-reduce) on the one GPU of the box.  `auto` must pick eager launches when a process group is alive (a captured
    step can abort the process through ProcessGroupNCCL's watchdog, bench.py main()), and the line must carry the
    `collectives` diagnostics a real scaling run will be read with."""
    env = dict(os.environ, VSPW_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0", VSPW_WATCHDOG_S="100")
    env.pop("WORLD_SIZE", None)
    env.pop("VSPW_BENCH_SHARED_GPU", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
           "--no-host-probe"]
    rc, stdout, err = _run(cmd, env, 150)
    assert rc == 0, err[-4000:]
    out = json.loads([ln for ln in stdout.splitlines() if ln.startswith("{")][-1])
    assert out["config"]["execution"] == "eager launches" and out["config"]["rccl_ranks"] == 1
    c = out["collectives"]
    assert c["syncbn_exchanges_per_step"] == 224            # 112 BatchNorm layers, forward + backward
    assert "peer exchange" in c["syncbn_exchange"]          # hipIpc arenas + one kernel per exchange, self-tested
    assert 0.2 < c["syncbn_exchange_ms_per_step"] < 50.0
    assert c["grad_buckets"] >= 8 and c["allreduce_exposed_ms_per_step"] >= 0.0
    assert c["rccl_graph_capture"] is None and c["sync_bn"] is True and c["sync_bn_formula"] == "var+eps"
    assert out["roofline"]["effective_direct_conv_tflops"] > out["roofline"]["achieved"]  # Winograd launches counted
