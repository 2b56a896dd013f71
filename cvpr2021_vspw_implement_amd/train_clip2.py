"""Training driver, mirroring the reference's train_clip2.py (same flags, same model dispatch, same feed_dict assembly,
same four SGD groups + poly schedule, same checkpoint files) on the HIP hot path.

Differences that come from the MI355X-native design, not from the semantics:
  * one process per GPU (`python -m torch.distributed.run --nproc-per-node N -m cvpr2021_vspw_implement_amd.train_clip2
    ...`) instead of a single process with nn.DataParallel over `--gpu_num` devices (train_clip2.py:359-364): every
    rank loads `--batchsize / world` clips per step, gradients are averaged over RCCL, BatchNorm statistics are
    synchronised (distributed.DataParallelOverRCCL);
  * the DataLoader workers only decode; flip / rescale / crop / normalise run on the GPU (dataset2.DeviceTransform);
  * checkpoints are always written with the `module.` key prefix the reference's multi-GPU runs produce (and that
    its loaders strip unconditionally, train_clip2.py:347-357, test_clip2.py:265-271); loading accepts both.
Methods outside SURVEY.md §8 (tdnet, ETC, propnet, our_warp*, etc_ocr) raise NotImplementedError, like an unknown
`--method` does in the reference (train_clip2.py:321).
"""
import argparse
import os
import random
import time
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn

from . import distributed as vdist
from . import optim as voptim
from .config import cfg
from .dataset2 import (BaseDataset_clip, BaseDataset_longclip, DeviceTransform, TestDataset_clip, collate_raw)
from .models import (ClipOCRNet, Clip_PSP, ModelBuilder, NetWarp, NetWarp_ocr, Non_local3d)
from .utils import AverageMeter, Evaluator, parse_devices, setup_logger

_OUT_OF_SCOPE = ("tdnet", "ETC", "our_warp", "propnet", "our_warp_merge", "etc_ocr")
METHODS = ["netwarp", "ETC", "nonlocal3d", "tdnet", "our_warp", "propnet", "our_warp_merge", "clip_psp", "clip_ocr",
           "netwarp_ocr", "etc_ocr"]


def build_module(cfg, args, num_class, training=True):
    """train_clip2.py:258-321 / test_clip2.py:213-261."""
    if args.method in _OUT_OF_SCOPE:
        raise NotImplementedError("--method %s is outside the hot path rebuilt here (SURVEY.md §8)" % args.method)
    enc = ModelBuilder.build_encoder(arch=cfg.MODEL.arch_encoder.lower(), fc_dim=cfg.MODEL.fc_dim,
                                     weights=cfg.MODEL.weights_encoder if training else "", args=args)
    dec = ModelBuilder.build_decoder(arch=cfg.MODEL.arch_decoder.lower(), fc_dim=cfg.MODEL.fc_dim, num_class=num_class,
                                     weights=cfg.MODEL.weights_decoder if training else "", use_softmax=not training)
    crit = nn.NLLLoss(ignore_index=255 if training else -1)
    ds = 0.4 if training else None
    if args.method == "netwarp":
        return NetWarp(enc, dec, crit, args, cfg.TRAIN.deep_sup_scale)
    if args.method == "nonlocal3d":
        return Non_local3d(args, enc, crit)
    if args.method == "clip_psp":
        return Clip_PSP(enc, crit, args, deep_sup_scale=ds)
    if args.method == "clip_ocr":
        return ClipOCRNet(enc, crit, args, deep_sup_scale=ds)
    if args.method == "netwarp_ocr":
        return NetWarp_ocr(enc, crit, args, deep_sup_scale=ds)
    raise NotImplementedError


def make_batch(args, clip_imgs, clip_gts, it_):
    """The feed_dict of train_clip2.py:46-83 from the T per-frame batch tensors."""
    clip_imgs, clip_gts = list(clip_imgs), list(clip_gts)
    batch_data = {}
    idx = args.clip_num / 2 if args.clip_num % 2 == 0 else (args.clip_num - 1) / 2
    if args.method == "nonlocal3d":
        batch_data["clipimgs_data"] = clip_imgs
        batch_data["cliplabels_data"] = clip_gts
    elif args.method in ("netwarp", "netwarp_ocr"):
        assert args.clip_num == 2
        assert args.dilation_num == 0
        batch_data["img_data"] = clip_imgs.pop(int(idx))
        batch_data["seg_label"] = clip_gts.pop(int(idx))
        batch_data["clipimgs_data"] = clip_imgs
        batch_data["cliplabels_data"] = clip_gts
    elif args.method in ("clip_psp", "clip_ocr"):
        batch_data["img_data"] = clip_imgs[0]
        batch_data["seg_label"] = clip_gts[0]
        batch_data["clipimgs_data"] = clip_imgs[1:]
        batch_data["cliplabels_data"] = clip_gts[1:]
    else:
        raise NotImplementedError
    batch_data["step"] = it_
    return batch_data


class GraphedTrainStep(object):
    """--hip_graph: the training step of `train` (zero_grad, forward, loss, backward, gradient all-reduce, SGD) captured
    once as a hipGraph over static copies of the batch tensors and replayed per iteration (graph.GraphedStep).  The
    capture needs warm-up executions of the step; parameters, buffers and momentum are snapshotted before and put back
    after them, so the run takes exactly the reference's sequence of updates - and, every kernel being order-independent,
    arrives at the same bits as the eager loop (tests/test_drivers_gpu.py).  Batches of another shape run eagerly."""

    def __init__(self, module, optimizers, args, clip_imgs, clip_gts, feed=None):
        """optimizers: one optimizer or several (train.py steps one per net); feed(imgs, gts) -> the feed_dict of one
        step from the static tensors (default: this driver's make_batch)."""
        from . import ops
        from .graph import GraphedStep

        self.imgs = [t.clone() for t in clip_imgs]
        self.gts = [t.clone() for t in clip_gts]
        opts = list(optimizers) if isinstance(optimizers, (list, tuple)) else [optimizers]
        self.optimizers = opts
        if feed is None:
            feed = lambda imgs, gts: make_batch(args, imgs, gts, 0)  # noqa: E731
        inner = module.module if hasattr(module, "module") else module
        snap = {k: v.detach().clone() for k, v in inner.state_dict().items()}
        had_momentum = {p: (o, "momentum_buffer" in o.state[p]) for o in opts for g in o.param_groups
                        for p in g["params"]}
        mom = {p: o.state[p]["momentum_buffer"].clone() for p, (o, h) in had_momentum.items() if h}

        def step():
            module.zero_grad()
            loss, acc = module(feed(self.imgs, self.gts))
            loss, acc = loss.mean(), acc.mean()
            loss.backward()
            if hasattr(module, "finish_gradients"):
                module.finish_gradients()
            for o in opts:
                o.step()
            return loss, acc

        try:
            self.graph = GraphedStep(step, warmup=2, stream=getattr(args, "_work_stream", None))
        finally:  # also when the capture fails (an RCCL build that cannot be captured): the warm-up steps are undone
            try:
                torch.cuda.synchronize()
            except RuntimeError:  # an invalidated capture can make the sync itself raise: the restore must still run
                pass
            with torch.no_grad():
                for k, v in inner.state_dict().items():
                    v.copy_(snap[k])
                for p, (o, h) in had_momentum.items():
                    if "momentum_buffer" in o.state[p]:
                        buf = o.state[p]["momentum_buffer"]
                        buf.copy_(mom[p]) if h else buf.zero_()  # zero-filled buffers = torch's first-step semantics
            ops.invalidate_inference_cache()

    def matches(self, clip_imgs, clip_gts):
        return (len(clip_imgs) == len(self.imgs) and all(a.shape == b.shape for a, b in zip(clip_imgs, self.imgs))
                and all(a.shape == b.shape and a.dtype == b.dtype for a, b in zip(clip_gts, self.gts)))

    def __call__(self, clip_imgs, clip_gts):
        for dst, src in zip(self.imgs + self.gts, list(clip_imgs) + list(clip_gts)):
            dst.copy_(src)
        for o in self.optimizers:
            o.set_lrs()
        return self.graph.replay()


def train(segmentation_module, data_loader, optimizers, history, epoch, cfg, args, transform=None, log=print):
    """One epoch: train_clip2.py:26-124."""
    batch_time, data_time = AverageMeter(), AverageMeter()
    ave_total_loss, ave_acc = AverageMeter(), AverageMeter()
    segmentation_module.train(not cfg.TRAIN.fix_bn)
    epoch_iters = len(data_loader)
    max_iters = epoch_iters * cfg.TRAIN.num_epoch
    tic = time.time()
    it_ = 0
    for i, data in enumerate(data_loader):
        it_ += 1
        clip_imgs, clip_gts = transform(data) if transform is not None else data
        batch_data = make_batch(args, clip_imgs, clip_gts, it_)
        data_time.update(time.time() - tic)
        cur_iter = i + (epoch - 1) * epoch_iters
        adjust_learning_rate(optimizers, cur_iter, cfg, max_iters, args)
        graphed = getattr(args, "_graphed_step", None)
        if getattr(args, "hip_graph", False) and graphed is None:
            try:
                graphed = args._graphed_step = GraphedTrainStep(segmentation_module, optimizers, args, clip_imgs,
                                                                clip_gts)
            except Exception as e:  # state was restored by GraphedTrainStep: carry on launch by launch
                log("hipGraph capture of the training step failed (%s: %s); running eagerly" % (type(e).__name__, e))
                args.hip_graph = False
        use_graph = graphed is not None and graphed.matches(clip_imgs, clip_gts)
        if not hasattr(args, "_hip_graph_requested"):  # (the flag itself is cleared on a rank whose capture failed)
            args._hip_graph_requested = bool(getattr(args, "hip_graph", False))
        if args._hip_graph_requested and vdist._world() > 1:
            # a rank that replays its graph while a peer runs this step launch by launch (a batch of another shape
            # there) would issue a different sequence of collectives: replay only when EVERY rank replays
            use_graph = vdist.all_agree(use_graph)
        if use_graph:
            loss, acc = graphed(clip_imgs, clip_gts)
        else:
            segmentation_module.zero_grad()
            loss, acc = segmentation_module(batch_data)
            loss = loss.mean()
            acc = acc.mean()
            loss.backward()
            if hasattr(segmentation_module, "finish_gradients"):
                segmentation_module.finish_gradients()  # wait for the bucketed RCCL all-reduce
            optimizers.step()
        batch_time.update(time.time() - tic)
        tic = time.time()
        loss_value, acc_value = loss.data.item(), acc.data.item()  # ONE device sync per step, paid here
        vdist.step_guard(segmentation_module, loss_value)  # collective with more than one rank: all raise or none
        ave_total_loss.update(loss_value)
        ave_acc.update(acc_value * 100)
        log("Epoch: [{}][{}/{}], Time: {:.2f}, Data: {:.2f}, lr_encoder: {:.6f}, lr_decoder: {:.6f}, "
            "Accuracy: {:4.2f}, Loss: {:.6f}".format(epoch, i, epoch_iters, batch_time.average(), data_time.average(),
                                                     cfg.TRAIN.running_lr_encoder, cfg.TRAIN.running_lr_decoder,
                                                     ave_acc.average(), ave_total_loss.average()))
        fractional_epoch = epoch - 1 + 1. * i / epoch_iters
        history["train"]["epoch"].append(fractional_epoch)
        history["train"]["loss"].append(loss_value)
        history["train"]["acc"].append(acc_value)


def test(segmentation_module, args, transform, log=print, rank=0, world=1):
    """Validation pass of train_clip2.py:126-172 (every 15th frame of the `val` videos).  With world > 1 the videos
    are sharded round-robin over the ranks and the confusion matrices summed (one all-reduce): every rank takes part,
    so no rank sits in a collective of the next epoch while rank 0 validates alone (RCCL watchdog), and the metrics
    are those of the whole split (confusion matrices add)."""
    segmentation_module.eval()
    evaluator = Evaluator(args.num_class)
    log("validation")
    with open(os.path.join(args.dataroot, "val.txt"), "r") as f:
        videolists = [line[:-1] for line in f.readlines()]
    for video in videolists[rank::world]:
        test_dataset = TestDataset_clip(args.dataroot, video, args, is_train=True)
        loader = torch.utils.data.DataLoader(test_dataset, batch_size=1, shuffle=False, num_workers=args.workers,
                                             drop_last=False, collate_fn=collate_raw)
        for i, data in enumerate(loader):
            imgs_all, gts_all = transform(data)
            imgs, gts, clip_imgs = imgs_all[0], gts_all[0], imgs_all[1:]
            batch_data = {"img_data": imgs, "seg_label": gts, "clipimgs_data": clip_imgs}
            segSize = (imgs.size(2), imgs.size(3))
            with torch.no_grad():
                scores = segmentation_module(batch_data, segSize=segSize)
                pred = torch.argmax(scores, dim=1).data.cpu().numpy()
                target = gts.squeeze(1).cpu().numpy()
                evaluator.add_batch(target, pred)
    if world > 1:
        cm = torch.from_numpy(evaluator.confusion_matrix).to(transform.device if hasattr(transform, "device")
                                                             else "cuda")
        vdist.all_reduce(cm)
        evaluator.confusion_matrix = cm.cpu().numpy()
    Acc = evaluator.Pixel_Accuracy()
    Acc_class = evaluator.Pixel_Accuracy_Class()
    mIoU = evaluator.Mean_Intersection_over_Union()
    FWIoU = evaluator.Frequency_Weighted_Intersection_over_Union()
    log("Validation:")
    log("Acc:{}, Acc_class:{}, mIoU:{}, fwIoU: {}".format(Acc, Acc_class, mIoU, FWIoU))
    return Acc, Acc_class, mIoU, FWIoU


def _with_module_prefix(sd):
    return OrderedDict((("module." + k) if not k.startswith("module.") else k, v) for k, v in sd.items())


def strip_module_prefix(sd):
    """The reference strips 7 characters unconditionally (train_clip2.py:350-353); only do it when they are there."""
    return OrderedDict(((k[7:] if k.startswith("module.") else k), v) for k, v in sd.items())


def checkpoint(opt, nets, history, args, epoch):
    """train_clip2.py:179-189: `<saveroot>/model_epoch_N.pth` and `opt_epoch_N.pth` (rank 0 only)."""
    if not vdist.dist.is_initialized() or vdist.dist.get_rank() == 0:
        print("Saving checkpoints...")
        if not os.path.exists(args.saveroot):
            os.makedirs(args.saveroot)
        mod = nets.module if hasattr(nets, "module") else nets
        torch.save(_with_module_prefix(mod.state_dict()), "{}/model_epoch_{}.pth".format(args.saveroot, epoch))
        torch.save(opt.state_dict(), "{}/opt_epoch_{}.pth".format(args.saveroot, epoch))
    vdist.checkpoint_barrier()  # every rank: nobody starts the next step while rank 0 is still writing


def create_optimizers(model, cfg, args):
    """train_clip2.py:215-236."""
    return voptim.create_optimizers(model, args.lr, weight_decay=cfg.TRAIN.weight_decay, momentum=cfg.TRAIN.beta1,
                                    fix=args.fix)


def adjust_learning_rate(optimizer, cur_iter, cfg, max_iters, args):
    """train_clip2.py:239-252 (only running_lr_encoder is updated there; running_lr_decoder keeps its start value)."""
    cfg.TRAIN.running_lr_encoder = voptim.adjust_learning_rate(optimizer, cur_iter, max_iters, args.lr,
                                                               lr_pow=cfg.TRAIN.lr_pow, fix=args.fix)


def main(cfg, gpus, args):
    rank, local_rank, world = vdist.init_from_env()
    device = torch.device("cuda", local_rank if world > 1 else args.start_gpu)
    torch.cuda.set_device(device)
    log = print if rank == 0 else (lambda *a, **k: None)
    seed = cfg.TRAIN.seed + rank
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)

    segmentation_module = build_module(cfg, args, args.num_class, training=True)
    if args.method in ("clip_psp", "clip_ocr"):
        dataset_train = BaseDataset_longclip(args, "train")
    else:
        dataset_train = BaseDataset_clip(args, "train")
    if args.batchsize % world:
        raise ValueError("--batchsize %d must be divisible by the number of ranks %d" % (args.batchsize, world))
    sampler = None
    if world > 1:
        sampler = torch.utils.data.distributed.DistributedSampler(dataset_train, num_replicas=world, rank=rank,
                                                                  shuffle=True, seed=cfg.TRAIN.seed, drop_last=True)
    loader_train = torch.utils.data.DataLoader(dataset_train, batch_size=args.batchsize // world,
                                               shuffle=sampler is None, sampler=sampler, num_workers=args.workers,
                                               drop_last=True, pin_memory=False, collate_fn=collate_raw)
    log("1 Epoch = {} iters".format(len(loader_train)))
    transform = DeviceTransform(device)

    segmentation_module.cuda(device)
    optimizer = create_optimizers(segmentation_module, cfg, args)
    if args.resume_epoch != 0:
        to_load = torch.load(os.path.join("./resume", "model_epoch_{}.pth".format(args.resume_epoch)),
                             map_location=device)
        cfg.TRAIN.start_epoch = args.resume_epoch
        segmentation_module.load_state_dict(strip_module_prefix(to_load))
        optimizer.load_state_dict(torch.load(os.path.join("./resume", "opt_epoch_{}.pth".format(args.resume_epoch)),
                                             map_location=device))
        log("resume from epoch {}".format(args.resume_epoch))
    if world > 1 and getattr(args, "hip_graph", False) and os.environ.get("VSPW_GRAPH_WITH_COLLECTIVES") != "1":
        # capturing RCCL collectives can abort the process through ProcessGroupNCCL's watchdog thread (bench.py main())
        log("--hip_graph is ignored with %d ranks (set VSPW_GRAPH_WITH_COLLECTIVES=1 to capture anyway)" % world)
        args.hip_graph = False
    if world > 1:
        # one non-default stream for hook registration, training and (with --hip_graph) capture: see graph.GraphedStep
        args._work_stream = torch.cuda.Stream(device)
        args._work_stream.wait_stream(torch.cuda.current_stream(device))
        torch.cuda.set_stream(args._work_stream)
        # SyncBN inverse standard deviation over several devices: the reference's multi-device path computes
        # clamp(var, eps)^-1/2 (models/sync_batchnorm/batchnorm.py:150), its single-device path (var + eps)^-1/2;
        # "reference" reproduces the former, "single" makes N ranks compute what ONE device would on the whole batch
        segmentation_module = vdist.DataParallelOverRCCL(
            segmentation_module, sync_bn_clamp_var=getattr(args, "syncbn_formula", "reference") == "reference")

    history = {"train": {"epoch": [], "loss": [], "acc": []}}
    for epoch in range(cfg.TRAIN.start_epoch, cfg.TRAIN.num_epoch):
        log("Epoch {}".format(epoch))
        if sampler is not None:
            sampler.set_epoch(epoch)
        train(segmentation_module, loader_train, optimizer, history, epoch + 1, cfg, args, transform, log)
        if (epoch + 1) % args.ckpt_every == 0:
            checkpoint(optimizer, segmentation_module, history, args, epoch + 1)
            if args.validation:
                test(segmentation_module.module if hasattr(segmentation_module, "module") else segmentation_module,
                     args, transform, log, rank, world)
    log("Training Done!")
    if hasattr(segmentation_module, "close"):
        segmentation_module.close()  # peer-exchange arenas / IPC mappings (collective over the ranks)
    return history


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected.")


def build_parser():
    """The flags of train_clip2.py:379-489 (same names, types, defaults)."""
    p = argparse.ArgumentParser(description="VSPW clip training on the MI355X hot path")
    p.add_argument("--cfg", default="config/ade20k-resnet50dilated-ppm_deepsup.yaml", metavar="FILE", type=str)
    p.add_argument("--gpus", default="0-3")
    p.add_argument("--predir", default="../../ade20k-hrnetv2-c1")
    p.add_argument("--num_class", type=int, default=124)
    p.add_argument("--batchsize", type=int, default=16)
    p.add_argument("--workers", type=int, default=0)
    p.add_argument("--start_gpu", type=int, default=0)
    p.add_argument("--gpu_num", type=int, default=1)
    p.add_argument("--dataroot", type=str, default="")
    p.add_argument("--trainfps", type=int, default=1)
    p.add_argument("--lr", type=float, default=0.02)
    p.add_argument("--multi_scale", type=str2bool, default=False)
    p.add_argument("--saveroot", type=str, default="")
    p.add_argument("--totalepoch", type=int, default=30)
    p.add_argument("--dataroot2", type=str, default="")
    p.add_argument("--usetwodata", type=str2bool, default=False)
    p.add_argument("--cropsize", type=int, default=531)
    p.add_argument("--validation", type=str2bool, default=True)
    p.add_argument("--lesslabel", type=str2bool, default=False)
    p.add_argument("--clip_num", type=int, default=5)
    p.add_argument("--dilation_num", type=int, default=3)
    p.add_argument("--clip_up", type=str2bool, default=False)
    p.add_argument("--clip_middle", type=str2bool, default=False)
    p.add_argument("--fix", type=str2bool, default=False)
    p.add_argument("--othergt", type=str2bool, default=False)
    p.add_argument("--propclip2", type=str2bool, default=False)
    p.add_argument("--early_usecat", type=str2bool, default=False)
    p.add_argument("--earlyfuse", type=str2bool, default=False)
    p.add_argument("--weight_decay", type=float, default=1e-4)
    p.add_argument("--allsup", type=str2bool, default=False)
    p.add_argument("--allsup_scale", type=float, default=0.3)
    p.add_argument("--deepsup_scale", type=float, default=0.4)
    p.add_argument("--linear_combine", type=str2bool, default=False)
    p.add_argument("--distsoftmax", type=str2bool, default=False)
    p.add_argument("--distnearest", type=str2bool, default=False)
    p.add_argument("--temp", type=float, default=3)
    p.add_argument("--max_distances", type=str, default="10")
    p.add_argument("--pre_enc", type=str, default="")
    p.add_argument("--pre_dec", type=str, default="")
    p.add_argument("--method", type=str, default="", choices=METHODS)
    p.add_argument("--dilation2", type=str, default="2,5,9")
    p.add_argument("--resume_epoch", type=int, default=0)
    p.add_argument("--clipocr_all", type=str2bool, default=False)
    p.add_argument("--use_memory", type=str2bool, default=False)
    p.add_argument("--memory_num", type=int, default=8)
    p.add_argument("--st_weight", type=float, default=0.1)
    p.add_argument("--psp_weight", type=str2bool, default=False)
    # additions (no reference counterpart): checkpoint period (the reference hard-codes 20, train_clip2.py:386) and the
    # RAFT checkpoint NetWarp loads (models/netwarp.py:72 hard-codes this path)
    p.add_argument("--ckpt_every", type=int, default=20)
    p.add_argument("--syncbn_formula", default="reference", choices=["reference", "single"],
                   help="several ranks: 'reference' = clamp(var, eps)^-1/2 like the reference's multi-GPU SyncBN "
                        "(batchnorm.py:150), 'single' = (var + eps)^-1/2 like one device on the full batch")
    p.add_argument("--hip_graph", action="store_true",
                   help="replay the training step as one captured hipGraph (fixed crop / batch shapes; no reference "
                        "counterpart)")
    p.add_argument("--raft_weights", type=str, default="./RAFT_core/raft-things.pth-no-zip")
    p.add_argument("opts", help="Modify config options using the command-line", default=None, nargs=argparse.REMAINDER)
    return p


def prepare(args, cfg):
    """train_clip2.py:489-528: merge the config, derive the schedule fields the loop reads."""
    args.max_distances = [int(dd) for dd in str(args.max_distances).split(",")]
    cfg.merge_from_file(args.cfg)
    cfg.merge_from_list(args.opts)
    cfg.MODEL.weights_encoder = args.pre_enc
    cfg.MODEL.weights_decoder = args.pre_dec
    gpus = [int(x.replace("gpu", "")) for x in parse_devices(args.gpus)]
    cfg.TRAIN.num_epoch = args.totalepoch
    cfg.TRAIN.max_iters = cfg.TRAIN.epoch_iters * cfg.TRAIN.num_epoch
    cfg.TRAIN.weight_decay = args.weight_decay
    cfg.TRAIN.lr_encoder = args.lr
    cfg.TRAIN.lr_decoder = args.lr
    cfg.TRAIN.running_lr_encoder = cfg.TRAIN.lr_encoder
    cfg.TRAIN.running_lr_decoder = cfg.TRAIN.lr_decoder
    return gpus


if __name__ == "__main__":
    args = build_parser().parse_args()
    gpus = prepare(args, cfg)
    logger = setup_logger(distributed_rank=int(os.environ.get("RANK", "0")))
    logger.info("Loaded configuration file {}".format(args.cfg))
    logger.info("Running with config:\n{}".format(cfg))
    if not os.path.isdir(cfg.DIR):
        os.makedirs(cfg.DIR)
    with open(os.path.join(cfg.DIR, "config.yaml"), "w") as f:
        f.write("{}".format(cfg))
    print(args)
    main(cfg, gpus, args)
