"""Per-frame training driver, mirroring the reference's train.py (the entry point of scripts/run_psp.sh / run_ocr.sh:
BASELINE configs 1-2 - SegmentationModule(encoder, decoder) on single VSPW frames) on the HIP hot path: same flags and
defaults, same model construction from the yaml config, the decay / no-decay parameter partition with one SGD per
net, the poly schedule, the four checkpoint files per epoch, validation on every 15th frame of the val videos.

Differences that come from the MI355X-native design, not from the semantics (as in train_clip2.py of this package):
one process per GPU instead of nn.DataParallel over `--gpu_num` devices (train.py:305-311) - every rank loads
`--batchsize / world` samples per step, gradients averaged over RCCL, BatchNorm statistics synchronised; DataLoader
workers only decode, the pixel pipeline runs on the GPU (dataset2.DeviceTransform); `--use_float16` (torch.cuda.amp) is
refused: the hot path is fp32, like the numbers it is checked against.
"""
import argparse
import os
import random
import time

import numpy as np
import torch
import torch.nn as nn

from . import distributed as vdist
from . import optim as voptim
from .config import cfg
from .dataset2 import BaseDataset, BaseDataset_longclip, DeviceTransform, collate_raw
from .models import ModelBuilder, SegmentationModule
from .train_clip2 import GraphedTrainStep, str2bool
from .utils import AverageMeter, Evaluator, parse_devices, setup_logger


def feed(args, data, transform, it_):
    """train.py:39-56: the clip dataset's frames are independent images of one batch (frame-major, as torch.cat over the
    per-frame batch tensors orders them); the per-frame dataset's batch is used as it is."""
    imgs, gts = transform(data, frames_as_batch=True) if args.use_clipdataset else transform(data)
    return {"img_data": imgs[0], "seg_label": gts[0], "step": it_}


def train(segmentation_module, data_loader, optimizers, history, epoch, cfg, args, transform, log=print):
    """One epoch: train.py:23-113."""
    batch_time, data_time = AverageMeter(), AverageMeter()
    ave_total_loss, ave_acc = AverageMeter(), AverageMeter()
    segmentation_module.train(not cfg.TRAIN.fix_bn)
    epoch_iters = len(data_loader)
    max_iters = epoch_iters * cfg.TRAIN.num_epoch
    tic = time.time()
    it_ = 0
    for i, data in enumerate(data_loader):
        it_ += 1
        batch_data = feed(args, data, transform, it_)
        data_time.update(time.time() - tic)
        adjust_learning_rate(optimizers, i + (epoch - 1) * epoch_iters, cfg, max_iters)
        imgs, gts = [batch_data["img_data"]], [batch_data["seg_label"]]
        graphed = getattr(args, "_graphed_step", None)
        if getattr(args, "hip_graph", False) and graphed is None:
            try:  # the whole step replayed as one hipGraph (train_clip2.GraphedTrainStep): a per-frame step is short
                # enough for the host to be the bottleneck when it issues ~1 000 launches from Python
                graphed = args._graphed_step = GraphedTrainStep(
                    segmentation_module, optimizers, args, imgs, gts,
                    feed=lambda im, gt: {"img_data": im[0], "seg_label": gt[0], "step": 0})
            except Exception as e:  # state was restored by GraphedTrainStep: carry on launch by launch
                log("hipGraph capture of the training step failed (%s: %s); running eagerly" % (type(e).__name__, e))
                args.hip_graph = False
        use_graph = graphed is not None and graphed.matches(imgs, gts)
        if not hasattr(args, "_hip_graph_requested"):  # (the flag itself is cleared on a rank whose capture failed)
            args._hip_graph_requested = bool(getattr(args, "hip_graph", False))
        if args._hip_graph_requested and vdist._world() > 1:
            # a rank that replays its graph while a peer runs this step launch by launch (a batch of another shape
            # there) would issue a different sequence of collectives: replay only when EVERY rank replays
            use_graph = vdist.all_agree(use_graph)
        if use_graph:
            loss, acc = graphed(imgs, gts)
        else:
            segmentation_module.zero_grad()
            loss, acc = segmentation_module(batch_data)
            loss, acc = loss.mean(), acc.mean()
            loss.backward()
            if hasattr(segmentation_module, "finish_gradients"):
                segmentation_module.finish_gradients()  # wait for the bucketed RCCL all-reduce
            for optimizer in optimizers:
                optimizer.step()
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
        history["train"]["epoch"].append(epoch - 1 + 1. * i / epoch_iters)
        history["train"]["loss"].append(loss_value)
        history["train"]["acc"].append(acc_value)


def test(segmentation_module, loader, args, transform, log=print, world=1):
    """Validation pass of train.py:115-160 over `loader` (BaseDataset 'val': whole frames).  With world > 1 every rank
    evaluates its shard of the loader and the confusion matrices are summed."""
    segmentation_module.eval()
    evaluator = Evaluator(42 if args.lesslabel else args.num_class)
    log("validation")
    for i, data in enumerate(loader):
        log("[{}]/[{}]".format(i, len(loader)))
        imgs, gts = transform(data)
        imgs, gts = imgs[0], gts[0]
        with torch.no_grad():
            scores = segmentation_module({"img_data": imgs, "seg_label": gts}, segSize=(imgs.size(2), imgs.size(3)))
            pred = torch.argmax(scores, dim=1).data.cpu().numpy()
            evaluator.add_batch(gts.squeeze(1).cpu().numpy(), pred)
    if world > 1:
        cm = torch.from_numpy(evaluator.confusion_matrix).to(transform.device)
        vdist.all_reduce(cm)
        evaluator.confusion_matrix = cm.cpu().numpy()
    Acc, Acc_class = evaluator.Pixel_Accuracy(), evaluator.Pixel_Accuracy_Class()
    mIoU, FWIoU = evaluator.Mean_Intersection_over_Union(), evaluator.Frequency_Weighted_Intersection_over_Union()
    log("Validation:")
    log("Acc:{}, Acc_class:{}, mIoU:{}, fwIoU: {}".format(Acc, Acc_class, mIoU, FWIoU))
    return Acc, Acc_class, mIoU, FWIoU


def checkpoint(nets, optimizers, history, args, epoch):
    """train.py:167-188: encoder / decoder weights and both optimizers, one file each (rank 0 only)."""
    if not vdist.dist.is_initialized() or vdist.dist.get_rank() == 0:
        print("Saving checkpoints...")
        net_encoder, net_decoder, _crit = nets
        if not os.path.exists(args.saveroot):
            os.makedirs(args.saveroot)
        torch.save(net_encoder.state_dict(), "{}/encoder_epoch_{}.pth".format(args.saveroot, epoch))
        torch.save(net_decoder.state_dict(), "{}/decoder_epoch_{}.pth".format(args.saveroot, epoch))
        optimizer_encoder, optimizer_decoder = optimizers
        torch.save(optimizer_encoder.state_dict(), "{}/opt_encoder_epoch_{}.pth".format(args.saveroot, epoch))
        torch.save(optimizer_decoder.state_dict(), "{}/opt_decoder_epoch_{}.pth".format(args.saveroot, epoch))
    vdist.checkpoint_barrier()  # every rank: nobody starts the next step while rank 0 is still writing


def group_weight(module):
    """train.py:191-211: weights of Linear / convolution modules decay, their biases and every BatchNorm parameter do
    not; every parameter must fall in one of the two."""
    decay, no_decay = [], []
    for m in module.modules():
        if isinstance(m, (nn.Linear, nn.modules.conv._ConvNd)):
            decay.append(m.weight)
            if m.bias is not None:
                no_decay.append(m.bias)
        elif isinstance(m, nn.modules.batchnorm._BatchNorm):
            no_decay.extend(p for p in (m.weight, m.bias) if p is not None)
    assert len(list(module.parameters())) == len(decay) + len(no_decay)
    return [dict(params=decay), dict(params=no_decay, weight_decay=.0)]


def create_optimizers(nets, cfg):
    """train.py:214-226: one SGD per net (the fused HIP SGD of optim.py, torch.optim.SGD semantics)."""
    net_encoder, net_decoder, _crit = nets
    return tuple(voptim.SGD(group_weight(net), lr=lr, momentum=cfg.TRAIN.beta1, weight_decay=cfg.TRAIN.weight_decay)
                 for net, lr in ((net_encoder, cfg.TRAIN.lr_encoder), (net_decoder, cfg.TRAIN.lr_decoder)))


def adjust_learning_rate(optimizers, cur_iter, cfg, max_iters):
    """train.py:229-238: poly schedule on both optimizers."""
    scale = (1. - float(cur_iter) / max_iters) ** cfg.TRAIN.lr_pow
    cfg.TRAIN.running_lr_encoder = cfg.TRAIN.lr_encoder * scale
    cfg.TRAIN.running_lr_decoder = cfg.TRAIN.lr_decoder * scale
    for opt, lr in zip(optimizers, (cfg.TRAIN.running_lr_encoder, cfg.TRAIN.running_lr_decoder)):
        for group in opt.param_groups:
            group["lr"] = lr


def build_module(cfg, args):
    """train.py:250-271."""
    num_class = 42 if args.lesslabel else args.num_class
    net_encoder = ModelBuilder.build_encoder(arch=cfg.MODEL.arch_encoder.lower(), fc_dim=cfg.MODEL.fc_dim,
                                             weights=cfg.MODEL.weights_encoder)
    net_decoder = ModelBuilder.build_decoder(arch=cfg.MODEL.arch_decoder.lower(), fc_dim=cfg.MODEL.fc_dim,
                                             num_class=num_class, weights=cfg.MODEL.weights_decoder)
    crit = nn.NLLLoss(ignore_index=255)
    if cfg.MODEL.arch_decoder.endswith("deepsup"):
        module = SegmentationModule(net_encoder, net_decoder, crit, cfg.TRAIN.deep_sup_scale)
    else:
        module = SegmentationModule(net_encoder, net_decoder, crit)
    return module, (net_encoder, net_decoder, crit)


def main(cfg, gpus, args):
    if args.use_float16:
        raise NotImplementedError("--use_float16 (torch.cuda.amp): the MI355X hot path computes in fp32")
    rank, local_rank, world = vdist.init_from_env()
    device = torch.device("cuda", local_rank if world > 1 else args.start_gpu)
    torch.cuda.set_device(device)
    log = print if rank == 0 else (lambda *a, **k: None)
    seed = cfg.TRAIN.seed + rank
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)

    segmentation_module, nets = build_module(cfg, args)
    dataset_train = BaseDataset_longclip(args, "train") if args.use_clipdataset else BaseDataset(args, "train")
    if args.batchsize % world:
        raise ValueError("--batchsize %d must be divisible by the number of ranks %d" % (args.batchsize, world))
    sampler = val_sampler = None
    dataset_val = BaseDataset(args, "val")
    if world > 1:
        sampler = torch.utils.data.distributed.DistributedSampler(dataset_train, num_replicas=world, rank=rank,
                                                                  shuffle=True, seed=cfg.TRAIN.seed, drop_last=True)
        # every validation frame exactly once over the ranks (DistributedSampler would pad the split with repeats,
        # which the summed confusion matrices would then count twice)
        val_sampler = list(range(rank, len(dataset_val), world))
    loader_train = torch.utils.data.DataLoader(dataset_train, batch_size=args.batchsize // world,
                                               shuffle=sampler is None, sampler=sampler, num_workers=args.workers,
                                               drop_last=True, pin_memory=False, collate_fn=collate_raw)
    log("1 Epoch = {} iters".format(len(loader_train)))
    # validation frames keep their own size (videos differ): one frame per batch unless the split is uniform
    loader_val = torch.utils.data.DataLoader(dataset_val, batch_size=1, shuffle=False, sampler=val_sampler,
                                             num_workers=args.workers, collate_fn=collate_raw)
    transform = DeviceTransform(device)
    segmentation_module.cuda(device)
    optimizers = create_optimizers(nets, cfg)
    if world > 1 and getattr(args, "hip_graph", False) and os.environ.get("VSPW_GRAPH_WITH_COLLECTIVES") != "1":
        log("--hip_graph is ignored with %d ranks (set VSPW_GRAPH_WITH_COLLECTIVES=1 to capture anyway)" % world)
        args.hip_graph = False
    if world > 1:
        args._work_stream = torch.cuda.Stream(device)
        args._work_stream.wait_stream(torch.cuda.current_stream(device))
        torch.cuda.set_stream(args._work_stream)
        segmentation_module = vdist.DataParallelOverRCCL(
            segmentation_module, sync_bn_clamp_var=getattr(args, "syncbn_formula", "reference") == "reference")
    history = {"train": {"epoch": [], "loss": [], "acc": []}}
    for epoch in range(cfg.TRAIN.start_epoch, cfg.TRAIN.num_epoch):
        log("Epoch {}".format(epoch))
        if sampler is not None:
            sampler.set_epoch(epoch)
        train(segmentation_module, loader_train, optimizers, history, epoch + 1, cfg, args, transform, log)
        checkpoint(nets, optimizers, history, args, epoch + 1)
        if args.validation:
            test(segmentation_module.module if hasattr(segmentation_module, "module") else segmentation_module,
                 loader_val, args, transform, log, world)
    log("Training Done!")
    if hasattr(segmentation_module, "close"):
        segmentation_module.close()  # peer-exchange arenas / IPC mappings (collective over the ranks)
    return history


def build_parser():
    """The flags of train.py:346-399 (same names, types, defaults)."""
    p = argparse.ArgumentParser(description="PyTorch Semantic Segmentation Training")
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
    p.add_argument("--multi_scale", type=str2bool, default=True)
    p.add_argument("--saveroot", type=str, default="")
    p.add_argument("--totalepoch", type=int, default=30)
    p.add_argument("--dataroot2", type=str, default="")
    p.add_argument("--usetwodata", type=str2bool, default=False)
    p.add_argument("--cropsize", type=int, default=531)
    p.add_argument("--validation", type=str2bool, default=True)
    p.add_argument("--lesslabel", type=str2bool, default=False)
    p.add_argument("--train_filter", type=str2bool, default=False)
    p.add_argument("--weight_decay", type=float, default=1e-4)
    p.add_argument("--use_clipdataset", type=str2bool, default=False)
    p.add_argument("--dilation2", type=str, default="2,5,9")
    p.add_argument("--clip_num", type=int, default=4)
    p.add_argument("--dilation_num", type=int, default=0)
    p.add_argument("--use_float16", type=str2bool, default=False)
    p.add_argument("opts", help="Modify config options using the command-line", default=None, nargs=argparse.REMAINDER)
    return p


EXTRA_FLAGS = ("syncbn_formula", "hip_graph")  # additions without a reference counterpart (added in __main__)


def prepare(args, cfg):
    """train.py:401-444: merge the config, derive the schedule fields the loop reads."""
    cfg.merge_from_file(args.cfg)
    cfg.merge_from_list(args.opts)
    cfg.MODEL.weights_encoder = args.predir
    cfg.MODEL.weights_decoder = ""
    gpus = [int(x.replace("gpu", "")) for x in parse_devices(args.gpus)]
    cfg.TRAIN.num_epoch = args.totalepoch
    cfg.TRAIN.max_iters = cfg.TRAIN.epoch_iters * cfg.TRAIN.num_epoch
    cfg.TRAIN.weight_decay = args.weight_decay
    cfg.TRAIN.lr_encoder = cfg.TRAIN.lr_decoder = args.lr
    cfg.TRAIN.running_lr_encoder = cfg.TRAIN.running_lr_decoder = args.lr
    return gpus


if __name__ == "__main__":
    parser = build_parser()
    parser.add_argument("--syncbn_formula", default="reference", choices=["reference", "single"])
    parser.add_argument("--hip_graph", action="store_true",
                        help="replay the training step as one captured hipGraph (fixed crop / batch shapes)")
    args = parser.parse_args()
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
