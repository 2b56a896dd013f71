"""bench.py's N > 1 path end to end on a 1-GPU box: `python bench.py --gpus 2` launches two ranks itself
(torch.distributed.run, 127.0.0.1 rendezvous); in the VSPW_BENCH_SHARED_GPU test mode both share the device over gloo
(RCCL refuses two ranks on one GPU), everything else is the production path: parameter broadcast, SyncBN statistics
exchange per BatchNorm, bucketed gradient averaging, barrier + max-over-ranks timing, one JSON line from rank 0."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_launches_two_ranks_and_reports_once(dev):
    env = dict(os.environ, VSPW_BENCH_SHARED_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--no-cpu-baseline", "--no-host-probe", "--no-kernel-timing"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
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


def test_bench_refuses_more_ranks_than_devices(dev):
    import torch

    n = torch.cuda.device_count() + 1
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("VSPW_BENCH_SHARED_GPU", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1"], env=env,
                       cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "GPU(s) are visible" in r.stderr


def test_bench_reports_collective_diagnostics_with_a_one_rank_rccl_group(dev):
    """VSPW_FORCE_COLLECTIVES=1: the N > 1 code path (RCCL process group, SyncBN exchange per BatchNorm, bucketed
    all-reduce) on the one GPU of the box.  `auto` must pick eager launches when a process group is alive (a captured
    step can abort the process through ProcessGroupNCCL's watchdog, bench.py main()), and the line must carry the
    `collectives` diagnostics a real scaling run will be read with."""
    env = dict(os.environ, VSPW_FORCE_COLLECTIVES="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("VSPW_BENCH_SHARED_GPU", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
           "--no-host-probe"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["config"]["execution"] == "eager launches" and out["config"]["rccl_ranks"] == 1
    c = out["collectives"]
    assert c["syncbn_exchanges_per_step"] == 224            # 112 BatchNorm layers, forward + backward
    assert 0.5 < c["syncbn_exchange_ms_per_step"] < 50.0
    assert c["grad_buckets"] >= 8 and c["allreduce_exposed_ms_per_step"] >= 0.0
    assert c["rccl_graph_capture"] is None and c["sync_bn"] is True and c["sync_bn_formula"] == "var+eps"
    assert out["roofline"]["effective_direct_conv_tflops"] > out["roofline"]["achieved"]  # Winograd launches counted
