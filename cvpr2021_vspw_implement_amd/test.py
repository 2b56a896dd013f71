"""Per-frame evaluation driver, mirroring the reference's test.py (the "val..." step of scripts/run_psp.sh / run_ocr.sh):
SegmentationModule(encoder, decoder, use_softmax=True) over every frame of every video of a split - global and per-video
mIoU / fwIoU (utils.Evaluator), optional palette PNG dumps - on the HIP hot path, frames normalised on the GPU
(dataset2.DeviceTransform).  Same flags as test.py:186-209; `--load_en` / `--load_de` take the files train.py writes.
Single process, single GPU (`--start_gpu`), like the reference."""
import argparse
import os
import pickle as pkl

import torch
import torch.nn as nn
from PIL import Image

from .config import cfg
from .dataset2 import DeviceTransform, TestDataset, collate_raw
from .models import ModelBuilder, SegmentationModule
from .train_clip2 import str2bool
from .utils import Evaluator, setup_logger, vspw_palette

_palette = vspw_palette()


def test(segmentation_module, loader, gpu, args, evaluator, eval_video, video, transform):
    """test.py:50-80: arg-max of the softmax scores at frame size against the labels, both evaluators fed."""
    segmentation_module.eval()
    for data in loader:
        imgs, gts = transform(data)
        imgs, gts = imgs[0], gts[0]
        gtnames = [s.names for s in data]
        with torch.no_grad():
            scores = segmentation_module({"img_data": imgs, "seg_label": gts}, segSize=(imgs.size(2), imgs.size(3)))
            pred = torch.argmax(scores, dim=1).data.cpu().numpy()
            target = gts.squeeze(1).cpu().numpy()
        evaluator.add_batch(target, pred)
        eval_video.add_batch(target, pred)
        if args.is_save:
            out = os.path.join(args.saveroot, video)
            if not os.path.exists(out):
                os.makedirs(out)
            for j in range(pred.shape[0]):
                im = Image.fromarray(pred[j].astype("uint8")).convert("P")
                im.putpalette(_palette)
                im.save(os.path.join(out, gtnames[j]))


def build_module(cfg, args):
    """test.py:90-107."""
    net_encoder = ModelBuilder.build_encoder(arch=cfg.MODEL.arch_encoder, fc_dim=cfg.MODEL.fc_dim,
                                             weights=cfg.MODEL.weights_encoder)
    net_decoder = ModelBuilder.build_decoder(arch=cfg.MODEL.arch_decoder, fc_dim=cfg.MODEL.fc_dim,
                                             num_class=args.num_class, weights=cfg.MODEL.weights_decoder,
                                             use_softmax=True)
    return SegmentationModule(net_encoder, net_decoder, nn.NLLLoss(ignore_index=-1))


def main(cfg, gpu, args, log=print):
    torch.cuda.set_device(gpu)
    device = torch.device("cuda", args.start_gpu)
    segmentation_module = build_module(cfg, args)
    segmentation_module.cuda(device)
    transform = DeviceTransform(device)
    with open(os.path.join(args.dataroot, args.split + ".txt")) as f:
        videolists = [line[:-1] for line in f.readlines()]
    evaluator, eval_video = Evaluator(args.num_class), Evaluator(args.num_class)
    total_vmIOU = total_vfwIOU = 0.0
    per_video, names = [], []
    for video in videolists:
        eval_video.reset()
        dataset_test = TestDataset(args.dataroot, video, args)
        loader_test = torch.utils.data.DataLoader(dataset_test, batch_size=args.batchsize, shuffle=False,
                                                  num_workers=getattr(args, "workers", 5), drop_last=False,
                                                  collate_fn=collate_raw)
        test(segmentation_module, loader_test, gpu, args, evaluator, eval_video, video, transform)
        v_mIOU = eval_video.Mean_Intersection_over_Union()
        per_video.append(v_mIOU)
        names.append(video)
        log(video, v_mIOU)
        total_vmIOU += v_mIOU
        total_vfwIOU += eval_video.Frequency_Weighted_Intersection_over_Union()
    if getattr(args, "dump_video_miou", True):
        with open("vmiou_hr.pkl", "wb") as f:  # (test.py:148-149 dumps [v, n]; its v holds the list itself - a slip)
            pkl.dump([per_video, names], f)
    total_vmIOU /= len(videolists)
    total_vfwIOU /= len(videolists)
    Acc, Acc_class = evaluator.Pixel_Accuracy(), evaluator.Pixel_Accuracy_Class()
    mIoU, FWIoU = evaluator.Mean_Intersection_over_Union(), evaluator.Frequency_Weighted_Intersection_over_Union()
    log("Acc:{}, Acc_class:{}, mIoU:{}, fwIoU: {}, video mIOU: {}, video fwIOU: {}".format(
        Acc, Acc_class, mIoU, FWIoU, total_vmIOU, total_vfwIOU))
    log("Inference done!")
    return {"Acc": Acc, "Acc_class": Acc_class, "mIoU": mIoU, "fwIoU": FWIoU, "video_mIoU": total_vmIOU,
            "video_fwIoU": total_vfwIOU}


def build_parser():
    """The flags of test.py:186-209 (same names, types, defaults)."""
    p = argparse.ArgumentParser(description="PyTorch Semantic Segmentation Testing")
    p.add_argument("--cfg", default="config/ade20k-hrnetv2.yaml", metavar="FILE", type=str)
    p.add_argument("opts", help="Modify config options using the command-line", default=None, nargs=argparse.REMAINDER)
    p.add_argument("--start_gpu", type=int, default=0)
    p.add_argument("--num_class", type=int, default=124)
    p.add_argument("--dataroot", type=str, default="")
    p.add_argument("--saveroot", type=str, default="")
    p.add_argument("--load_en", type=str, default="")
    p.add_argument("--load_de", type=str, default="")
    p.add_argument("--batchsize", type=int, default=4)
    p.add_argument("--split", type=str, default="val")
    p.add_argument("--is_save", type=str2bool, default=False)
    p.add_argument("--lesslabel", type=str2bool, default=False)
    p.add_argument("--use_720p", type=str2bool, default=False)
    return p


def prepare(args, cfg):
    """test.py:211-230."""
    cfg.merge_from_file(args.cfg)
    cfg.merge_from_list(args.opts)
    cfg.MODEL.arch_encoder = cfg.MODEL.arch_encoder.lower()
    cfg.MODEL.arch_decoder = cfg.MODEL.arch_decoder.lower()
    cfg.MODEL.weights_encoder = args.load_en
    cfg.MODEL.weights_decoder = args.load_de
    assert os.path.exists(cfg.MODEL.weights_encoder) and os.path.exists(cfg.MODEL.weights_decoder), \
        "checkpoint does not exitst!"
    if not os.path.isdir(args.saveroot):
        os.makedirs(args.saveroot)


if __name__ == "__main__":
    args = build_parser().parse_args()
    prepare(args, cfg)
    logger = setup_logger(distributed_rank=0)
    logger.info("Loaded configuration file {}".format(args.cfg))
    logger.info("Running with config:\n{}".format(cfg))
    main(cfg, args.start_gpu, args)
    print(args)
