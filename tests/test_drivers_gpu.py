"""End-to-end run of the driver mirrors on the tiny VSPW tree: train_clip2.main (decode -> device input pipeline ->
Clip_PSP / NetWarp step -> fused SGD -> checkpoint with the reference's key format) and test_clip2.main (checkpoint
load, per-video inference, Evaluator / video-consistency metrics)."""
import os

import numpy as np
import pytest
import torch

from oracle.det_data import make_tiny_vspw

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tree(tmp_path_factory):
    root = str(tmp_path_factory.mktemp("vspw_tiny"))
    make_tiny_vspw(root)
    return root


def _cfg(arch_dec):
    from cvpr2021_vspw_implement_amd.config import cfg

    c = cfg.clone()
    c.MODEL.arch_encoder = "resnet50dilated"
    c.MODEL.arch_decoder = arch_dec
    c.MODEL.fc_dim = 2048
    return c


def test_clip_psp_train_checkpoint_eval(dev, tree, tmp_path):
    import cvpr2021_vspw_implement_amd.test_clip2 as E
    import cvpr2021_vspw_implement_amd.train_clip2 as T

    save = str(tmp_path / "ck")
    args = T.build_parser().parse_args([
        "--method", "clip_psp", "--dataroot", tree, "--saveroot", save, "--batchsize", "3", "--cropsize", "40",
        "--clip_num", "4", "--dilation2", "3,6,9", "--totalepoch", "2", "--ckpt_every", "2", "--lr", "0.01",
        "--multi_scale", "true", "--workers", "0", "--gpus", "0"])
    cfg = _cfg("ppm_deepsup_clip")
    here = os.path.dirname(os.path.abspath(T.__file__))
    args.cfg = os.path.join(here, "config", "vsp-resnet101dilated-ppm_deepsup_clip.yaml")
    T.prepare(args, cfg)
    cfg.MODEL.arch_encoder = "resnet50dilated"
    lines = []
    hist = T.main(cfg, [0], args)
    losses = hist["train"]["loss"]
    assert len(losses) == 2 and all(np.isfinite(losses))           # 3 videos / batch 3 = 1 iteration per epoch
    sd = torch.load(os.path.join(save, "model_epoch_2.pth"), map_location="cpu")
    assert all(k.startswith("module.") for k in sd) and "module.encoder.conv1.weight" in sd
    assert os.path.exists(os.path.join(save, "opt_epoch_2.pth"))
    # evaluation driver on the checkpoint it wrote
    eargs = E.build_parser().parse_args([
        "--method", "clip_psp", "--dataroot", tree, "--split", "test", "--load", os.path.join(save, "model_epoch_2.pth"),
        "--batchsize", "1", "--clip_num", "4", "--dilation2", "3,6,9", "--vc_clip_num", "4", "--is_save", "true",
        "--saveroot", str(tmp_path / "pred")])
    eargs.max_distances = [10]
    out = E.main(_cfg("ppm_deepsup_clip"), 0, eargs, log=lambda *a: lines.append(a))
    assert 0.0 <= out["Acc"] <= 1.0 and 0.0 <= out["mIoU"] <= 1.0 and 0.0 <= out["video_mIoU"] <= 1.0
    assert np.isnan(out["VC"]) or 0.0 <= out["VC"] <= 1.0
    pngs = os.listdir(str(tmp_path / "pred" / "v_b"))
    assert len(pngs) == 9  # one palette PNG per frame of the video
    # resume (train_clip2.py:347-357 reads ./resume/model_epoch_N.pth + opt_epoch_N.pth relative to the cwd)
    import shutil

    os.makedirs(str(tmp_path / "run" / "resume"))
    for f in ("model_epoch_2.pth", "opt_epoch_2.pth"):
        shutil.copy(os.path.join(save, f), str(tmp_path / "run" / "resume" / f))
    cwd = os.getcwd()
    os.chdir(str(tmp_path / "run"))
    try:
        rargs = T.build_parser().parse_args([
            "--method", "clip_psp", "--dataroot", tree, "--saveroot", save, "--batchsize", "3", "--cropsize", "40",
            "--clip_num", "4", "--dilation2", "3,6,9", "--totalepoch", "3", "--ckpt_every", "3", "--lr", "0.01",
            "--resume_epoch", "2", "--validation", "false", "--gpus", "0"])
        rargs.cfg = args.cfg
        rcfg = _cfg("ppm_deepsup_clip")
        T.prepare(rargs, rcfg)
        rcfg.MODEL.arch_encoder = "resnet50dilated"
        rhist = T.main(rcfg, [0], rargs)
    finally:
        os.chdir(cwd)
    assert len(rhist["train"]["loss"]) == 1 and np.isfinite(rhist["train"]["loss"][0])  # only epoch 3 ran
    assert os.path.exists(os.path.join(save, "model_epoch_3.pth"))


def test_netwarp_train_step_with_hip_raft(dev, tree, tmp_path):
    """netwarp: clip_num 2, RAFT (random init here) -> FlowCNN -> warps, through the training driver; frames are
    padded to the 136-pixel crop so that RAFT's 1/8-resolution maps keep >= 16 rows."""
    import cvpr2021_vspw_implement_amd.train_clip2 as T

    args = T.build_parser().parse_args([
        "--method", "netwarp", "--dataroot", tree, "--saveroot", str(tmp_path / "ck2"), "--batchsize", "1",
        "--cropsize", "131", "--clip_num", "2", "--dilation_num", "0", "--totalepoch", "1", "--ckpt_every", "5",
        "--lr", "0.01", "--raft_weights", "", "--validation", "false"])
    cfg = _cfg("ppm_deepsup_clip")
    here = os.path.dirname(os.path.abspath(T.__file__))
    args.cfg = os.path.join(here, "config", "vsp-resnet101dilated-ppm_deepsup_clip.yaml")
    T.prepare(args, cfg)
    cfg.MODEL.arch_encoder = "resnet50dilated"
    hist = T.main(cfg, [0], args)
    assert len(hist["train"]["loss"]) == 3 and all(np.isfinite(hist["train"]["loss"]))


def test_hip_graph_training_loop_equals_the_eager_loop(dev, tree, tmp_path):
    """train_clip2.train with --hip_graph (one captured hipGraph replayed per iteration over static batch buffers)
    against the plain loop: same batches, same poly-LR schedule -> the same loss trace and bit-identical parameters,
    buffers and momentum (Dropout2d disabled: warm-up executions of the capture advance the Philox offset, so masks
    would differ; everything else is order-independent).  Also: the warm-up leaves no trace in the training state."""
    import cvpr2021_vspw_implement_amd.train_clip2 as T
    from helpers import load_det, zero_dropout

    def run(hip_graph):
        argv = ["--method", "clip_psp", "--dataroot", tree, "--saveroot", str(tmp_path / "g"), "--batchsize", "2",
                "--cropsize", "40", "--clip_num", "4", "--dilation2", "3,6,9", "--totalepoch", "1", "--lr", "0.01",
                "--workers", "0", "--gpus", "0"] + (["--hip_graph"] if hip_graph else [])
        args = T.build_parser().parse_args(argv)
        cfg = _cfg("ppm_deepsup_clip")
        here = os.path.dirname(os.path.abspath(T.__file__))
        args.cfg = os.path.join(here, "config", "vsp-resnet101dilated-ppm_deepsup_clip.yaml")
        T.prepare(args, cfg)
        cfg.MODEL.arch_encoder = "resnet50dilated"
        cfg.TRAIN.num_epoch = 1
        mod = T.build_module(cfg, args, args.num_class, training=True)
        load_det(mod)
        zero_dropout(mod)
        mod.to(dev)
        opt = T.create_optimizers(mod, cfg, args)
        g = torch.Generator().manual_seed(11)
        batches = []
        for _ in range(4):   # 4 iterations of B = 2 clips x (1 + 3) frames
            imgs = [torch.randn(2, 3, 40, 40, generator=g).to(dev) for _ in range(4)]
            gts = [torch.randint(0, args.num_class, (2, 1, 40, 40), generator=g).float().to(dev) for _ in range(4)]
            batches.append((imgs, gts))
        hist = {"train": {"epoch": [], "loss": [], "acc": []}}
        T.train(mod, batches, opt, hist, 1, cfg, args, transform=None, log=lambda *a: None)
        torch.cuda.synchronize()
        state = {k: v.detach().cpu().numpy() for k, v in mod.state_dict().items()}
        mom = [opt.state[p]["momentum_buffer"].cpu().numpy() for grp in opt.param_groups for p in grp["params"]
               if "momentum_buffer" in opt.state[p]]
        return hist["train"]["loss"], state, mom, getattr(args, "_graphed_step", None)

    l0, s0, m0, g0 = run(False)
    l1, s1, m1, g1 = run(True)
    assert g0 is None and g1 is not None
    assert len(l0) == 4 and l0 == l1, (l0, l1)
    for k in s0:
        assert np.array_equal(s0[k], s1[k]), k
    assert len(m0) == len(m1) and all(np.array_equal(a, b) for a, b in zip(m0, m1))


@pytest.mark.plumbing
def test_two_rank_training_driver(dev, tree, tmp_path):
    """train_clip2.main with WORLD_SIZE = 2 (torch.distributed.run; both ranks share the device over gloo in the
    VSPW_SHARED_GPU_TEST mode - RCCL refuses two ranks on one GPU): DistributedSampler shards, parameter broadcast,
    SyncBN + bucketed gradient averaging keep the replicas IDENTICAL, validation is sharded over the ranks with the
    confusion matrices all-reduced (no rank left waiting in a collective), rank 0 alone writes the checkpoint."""
    import subprocess
    import sys

    save = str(tmp_path / "ck2r")
    os.makedirs(save)
    env = dict(os.environ, VSPW_SHARED_GPU_TEST="1", HSA_ENABLE_IPC_MODE_LEGACY="0", VSPW_DIST_TIMEOUT_S="90",
               VSPW_WATCHDOG_S="100")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    here = os.path.dirname(os.path.abspath(__file__))
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(here, "two_rank_train_worker.py"), tree, save]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert r.stdout.count("Training Done!") == 1 and r.stdout.count("Saving checkpoints...") == 1   # rank 0 only
    assert r.stdout.count("Validation:") == 1 and "mIoU" in r.stdout
    assert os.path.exists(os.path.join(save, "model_epoch_2.pth"))
    d0, d1 = (np.load(os.path.join(save, "rank%d_digest.npy" % k)) for k in (0, 1))
    assert np.array_equal(d0, d1)                           # replicas in sync after two epochs (params AND BN buffers)
    l0, l1 = (np.load(os.path.join(save, "rank%d_loss.npy" % k)) for k in (0, 1))
    assert len(l0) == len(l1) == 2 and np.all(np.isfinite(l0)) and np.all(np.isfinite(l1))
    assert not np.array_equal(l0, l1)                       # each rank trained on its own shard


@pytest.mark.parametrize("hip_graph", [False, True])
@pytest.mark.parametrize("kind", ["clip_psp", "clip_ocr"])
def test_tcb_training_trajectory_follows_the_reference(dev, tmp_path, kind, hip_graph):
    """Twenty optimisation steps of the TCB heads through THIS repo's train_clip2.train (eager and --hip_graph) against the
    reference's own train() (train_clip2.py:26-124 with its create_optimizers / adjust_learning_rate: four SGD groups,
    head at 10x the encoder rate, duplicate parameter listings with PyTorch 1.3.1's in-place weight decay, poly schedule,
    BatchNorm running statistics over steps) run in the build container in float64, float32 and as a six-member float32
    ensemble whose first image carries a one-ulp perturbation (tests/golden/make_golden_trajectory.py ->
    tcb_train_trajectory_<kind>.npz; R50, T = 3, B = 2, 97 x 97, well-conditioned "damped" weights).
    Gate: at EVERY step |loss - float64| within 3 x the ensemble's own largest deviation at that step (floor 3e-6
    relative; 5 x once the ensemble itself has spread beyond 1e-3 - TCB-OCR after three updates) - i.e. the HIP
    trajectory is indistinguishable from a float32 realisation of the reference's loop.  The gate
    has teeth: the same loop with torch 2.10's out-of-place weight decay ("n32" in the fixture) leaves the envelope at
    step 1 by two orders of magnitude.  Final parameter norms per SGD group, momentum norms and the running statistics
    of five BatchNorm layers are held to the same ensemble yardstick."""
    import cvpr2021_vspw_implement_amd.train_clip2 as T
    from helpers import K, golden, load_det, zero_dropout
    from oracle.det_init import damp_residual_gammas, det_input, det_labels

    fx = golden("tcb_train_trajectory_" + kind)
    steps, num_epoch, Tn, B, S = (int(v) for v in fx["meta"])
    tag = "tcb_train_trajectory_" + kind
    argv = ["--method", kind, "--dataroot", str(tmp_path), "--saveroot", str(tmp_path / "t"), "--batchsize", str(B),
            "--cropsize", str(S), "--clip_num", str(Tn), "--dilation2", ",".join(str(3 * (i + 1)) for i in range(Tn - 1)),
            "--totalepoch", str(num_epoch), "--lr", repr(float(fx["lr"])), "--workers", "0", "--gpus", "0"] + \
           (["--hip_graph"] if hip_graph else [])
    args = T.build_parser().parse_args(argv)
    cfg = _cfg("ppm_deepsup_clip")  # (the per-frame decoder build_module constructs is unused by both TCB heads)
    here = os.path.dirname(os.path.abspath(T.__file__))
    args.cfg = os.path.join(here, "config", "vsp-resnet101dilated-ppm_deepsup_clip.yaml")
    T.prepare(args, cfg)
    cfg.MODEL.arch_encoder = "resnet50dilated"
    cfg.TRAIN.num_epoch, cfg.TRAIN.fix_bn, cfg.TRAIN.weight_decay = num_epoch, False, 1e-4
    mod = T.build_module(cfg, args, K, training=True)
    load_det(mod)
    sd = mod.state_dict()
    assert damp_residual_gammas(sd)
    mod.load_state_dict(sd)
    zero_dropout(mod)
    mod.to(dev)
    opt = T.create_optimizers(mod, cfg, args)
    assert [sum(g["mult"]) for g in opt.param_groups] == [int(v) for v in fx["group_sizes"]]
    assert [len(g["params"]) for g in opt.param_groups] == [int(v) for v in fx["group_unique"]]
    batches = []
    for it in range(steps):
        imgs = [torch.from_numpy(det_input("%s:img:%d:%d" % (tag, it, t), (B, 3, S, S))).to(dev) for t in range(Tn)]
        labs = [torch.from_numpy(det_labels("%s:lab:%d:%d" % (tag, it, t), (B, 1, S, S), K)).to(dev) for t in range(Tn)]
        batches.append((imgs, labs))
    hist = {"train": {"epoch": [], "loss": [], "acc": []}}
    T.train(mod, batches, opt, hist, 1, cfg, args, transform=None, log=lambda *a: None)
    torch.cuda.synchronize()
    assert (getattr(args, "_graphed_step", None) is not None) == hip_graph
    members = ("f32", "p0", "p1", "p2", "p3", "p4", "p5")
    l64 = fx["f64:loss"]
    ens = np.stack([np.abs(fx[m + ":loss"] - l64) for m in members]).max(0)
    err = np.abs(np.array(hist["train"]["loss"]) - l64)
    print(kind, "graph" if hip_graph else "eager", "|hip - ref64| / ensemble envelope per step:",
          " ".join("%.2f" % (e / max(g, 1e-12)) for e, g in zip(err, ens)))
    print("   |hip - ref64|", " ".join("%.1e" % e for e in err))
    print("   envelope     ", " ".join("%.1e" % e for e in ens))
    print("   torch-2.10-SGD run |n32 - ref64|", " ".join("%.1e" % e for e in np.abs(fx["n32:loss"] - l64)))
    # the yardstick is the MAXIMUM over only seven float32 realisations: one more realisation - HIP's, which itself changes
    # with any one-ulp difference in a weight transform - lands up to 2.6x outside it (measured over six orderings of these
    # cases, tools/diag/traj_order.py).  Gate: 3x where the seven still agree to 1e-3, 5x beyond (TCB-OCR from step 3 on:
    # the ensemble itself is 1e-2 apart): "no further out than a float32 realisation of the same loop"
    chaotic = ens > 1e-3
    for t in range(steps):
        gate = max(3e-6 * abs(l64[t]), (5.0 if chaotic[t] else 3.0) * ens[t])
        assert err[t] <= gate, (t, hist["train"]["loss"][t], float(l64[t]), float(err[t]), gate)
    fin = 6.0 if chaotic[-1] else 3.0  # the same for the end-of-run statistics below
    assert err[0] <= 3e-6 * abs(l64[0])  # before any update: the forward pass alone
    # the yardstick has teeth: the out-of-place-decay SGD would fail this very gate early on
    assert np.abs(fx["n32:loss"] - l64)[1] > 5.0 * max(3e-6 * abs(l64[1]), 3.0 * ens[1])
    acc = np.array(hist["train"]["acc"])
    assert np.abs(acc - fx["f64:acc"]).max() <= max(2.0 * np.stack([np.abs(fx[m + ":acc"] - fx["f64:acc"]) for m in members]).max(), 2e-4)

    def yard(key, got, floor):
        ref = fx["f64:" + key]
        scale = np.maximum(np.abs(ref), 1e-3 * np.abs(ref).max())
        gap = np.stack([np.abs(fx[m + ":" + key] - ref) for m in members]).max(0) / scale
        e = np.abs(got - ref) / scale
        assert np.median(e) <= max(fin * np.median(gap), floor), (key, float(np.median(e)), float(np.median(gap)))
        assert e.max() <= max(fin * gap.max(), 30 * floor), (key, float(e.max()), float(gap.max()), int(e.argmax()))

    names = [str(n) for n in fx["param_names"]]
    got = dict((k, float(p.detach().double().norm())) for k, p in mod.named_parameters())
    yard("param_norms", np.array([got[k] for k in names]), 1e-6)
    yard("group_norms", np.array([float(torch.sqrt(sum((p.detach().double() ** 2).sum() for p in g["params"])))
                                  for g in opt.param_groups]), 1e-6)
    seen, mom = set(), []
    for g in opt.param_groups:  # first-occurrence order over the groups, as the fixture lists them
        for p in g["params"]:
            if id(p) not in seen:
                seen.add(id(p))
                mom.append(float(opt.state[p]["momentum_buffer"].double().norm()))
    yard("momentum_norms", np.array(mom), 1e-5)
    state = mod.state_dict()
    for bn in (str(b) for b in fx["bn_layers"]):
        for what, key in (("rm", "running_mean"), ("rv", "running_var")):
            yard("%s:%s" % (what, bn), state["%s.%s" % (bn, key)].double().cpu().numpy(), 1e-6)
