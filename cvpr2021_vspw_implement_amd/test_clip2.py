"""Evaluation driver, mirroring the reference's test_clip2.py: per-video inference with the clip test datasets, global
and per-video mIoU / fwIoU (utils.Evaluator), the video-consistency score VC_n (utils.get_common), optional palette
PNG dumps - on the HIP hot path, decoded frames normalised on the GPU (dataset2.DeviceTransform).

Same flags as test_clip2.py:352-401.  `--load` takes a checkpoint written by either driver (keys with or without the
`module.` prefix, test_clip2.py:265-271).  Single process, single GPU (`--start_gpu`), as the reference's inference is
in practice (its DataParallel branch only replicates the module)."""
import argparse
import os

import numpy as np
import torch
from PIL import Image

from .config import cfg
from .dataset2 import DeviceTransform, TestDataset_clip, TestDataset_longclip, collate_raw
from .train_clip2 import METHODS, build_module, str2bool, strip_module_prefix
from .utils import Evaluator, get_common, setup_logger, vspw_palette

_palette = vspw_palette()


def _save(pred, args, video, name):
    imgpred = Image.fromarray(pred.astype("uint8")).convert("P")
    imgpred.putpalette(_palette)
    out = os.path.join(args.saveroot, video)
    if not os.path.exists(out):
        os.makedirs(out)
    imgpred.save(os.path.join(out, name.split(".")[0] + ".png"))


def _names(data):
    """Per-sample frame names of a collated raw batch (the reference's default collate yields a list of strings)."""
    return [s.names for s in data]


def test(segmentation_module, loader, gpu, args, evaluator, eval_video, video, transform):
    """test_clip2.py:28-87: one prediction per frame, the target frame plus its clip as context."""
    segmentation_module.eval()
    gtlist_, predlist_ = [], []
    h = w = 0
    for i, data in enumerate(loader):
        imgs_all, gts_all = transform(data)
        imgs, gts, clip_imgs = imgs_all[0], gts_all[0], imgs_all[1:]
        gtnames = _names(data)
        _, _, h, w = imgs.size()
        batch_data = {"img_data": imgs, "seg_label": gts, "clipimgs_data": clip_imgs}
        if args.use_memory:
            batch_data["is_clean_memory"] = i == 0
        segSize = (imgs.size(2), imgs.size(3))
        with torch.no_grad():
            scores = segmentation_module(batch_data, segSize=segSize)
            pred = torch.argmax(scores, dim=1).data.cpu().numpy()
            target = gts.squeeze(1).cpu().numpy()
            evaluator.add_batch(target, pred)
            eval_video.add_batch(target, pred)
            for jj in range(pred.shape[0]):
                predlist_.append(pred[jj])
                gtlist_.append(target[jj])
            if args.is_save:
                for j in range(pred.shape[0]):
                    _save(pred[j], args, video, gtnames[j])
    return gtlist_, predlist_, h, w


def test_all(segmentation_module, loader, gpu, args, evaluator, eval_video, video, transform):
    """test_clip2.py:88-190 (nonlocal3d): every clip scores all of its frames; a frame's scores are averaged once it has
    been seen clip_num times (or at the end of the video)."""
    segmentation_module.eval()
    gtlist_, predlist_ = [], []
    target_dic, pred_dic, nn_done = {}, {}, []
    h = w = 0

    def flush(nn_, tmp):
        pred = torch.argmax(tmp, dim=1).data.cpu().numpy()
        target = target_dic[nn_].squeeze(1).cpu().numpy()
        evaluator.add_batch(target, pred)
        eval_video.add_batch(target, pred)
        for jj in range(pred.shape[0]):
            predlist_.append(pred[jj])
            gtlist_.append(target[jj])
        if args.is_save:
            for j in range(pred.shape[0]):
                _save(pred[j], args, video, nn_)

    for i, data in enumerate(loader):
        imgs_all, gts_all = transform(data)
        imgs = imgs_all[0]
        clip_imgs, clip_targets = imgs_all[1:], gts_all[1:]
        gtnames = [[s.names[t] for s in data] for t in range(len(clip_imgs))]
        _, _, h, w = imgs.size()
        batch_data = {"clipimgs_data": clip_imgs, "cliplabels_data": clip_targets}
        segSize = (imgs.size(2), imgs.size(3))
        with torch.no_grad():
            scores = segmentation_module(batch_data, segSize=segSize)
            for score, clip_target, gtname in zip(scores, clip_targets, gtnames):
                for ii in range(score.size(0)):
                    ss, ll, nn_ = score[ii], clip_target[ii], gtname[ii]
                    if nn_ in nn_done:
                        continue
                    if nn_ not in target_dic:
                        target_dic[nn_] = ll.unsqueeze(0)
                    pred_dic.setdefault(nn_, []).append(ss.unsqueeze(0))
                    if len(pred_dic[nn_]) > args.clip_num - 1:
                        flush(nn_, torch.cat(pred_dic[nn_], dim=0).mean(dim=0, keepdim=True))
                        del pred_dic[nn_]
                        nn_done.append(nn_)
    for k, v in pred_dic.items():
        flush(k, torch.cat(v, dim=0).mean(dim=0, keepdim=True))
    return gtlist_, predlist_, h, w


def main(cfg, gpu, args, log=print):
    """test_clip2.py:198-333."""
    num_class = 42 if args.lesslabel else args.num_class
    device = torch.device("cuda", args.start_gpu)
    torch.cuda.set_device(device)
    segmentation_module = build_module(cfg, args, num_class, training=False)
    segmentation_module.cuda(device)
    if args.load:
        to_load = torch.load(args.load, map_location=device)
        segmentation_module.load_state_dict(strip_module_prefix(to_load))
    with open(os.path.join(args.dataroot, args.split + ".txt")) as f:
        videolists = [line[:-1] for line in f.readlines()]
    transform = DeviceTransform(device)
    evaluator, eval_video = Evaluator(num_class), Evaluator(num_class)
    evaluator.reset()
    eval_video.reset()
    total_vmIOU = total_vfwIOU = 0.0
    total_VC_acc = []
    for video in videolists:
        eval_video.reset()
        if args.method in ("clip_psp", "clip_ocr"):
            test_dataset = TestDataset_longclip(args.dataroot, video, args, is_train=False)
        else:
            test_dataset = TestDataset_clip(args.dataroot, video, args, is_train=False)
        loader_test = torch.utils.data.DataLoader(test_dataset, batch_size=args.batchsize, shuffle=False, num_workers=0,
                                                  drop_last=False, collate_fn=collate_raw)
        run = test_all if args.method == "nonlocal3d" else test
        gtlist_, predlist_, h, w = run(segmentation_module, loader_test, gpu, args, evaluator, eval_video, video,
                                       transform)
        accs = get_common(gtlist_, predlist_, args.vc_clip_num, h, w)
        if len(accs):
            log(sum(accs) / len(accs))
        total_VC_acc.extend(accs)
        v_mIOU = eval_video.Mean_Intersection_over_Union()
        total_vmIOU += v_mIOU
        total_vfwIOU += eval_video.Frequency_Weighted_Intersection_over_Union()
        log(video, v_mIOU)
    total_video = len(videolists)
    total_vmIOU, total_vfwIOU = total_vmIOU / total_video, total_vfwIOU / total_video
    Acc = evaluator.Pixel_Accuracy()
    Acc_class = evaluator.Pixel_Accuracy_Class()
    mIoU = evaluator.Mean_Intersection_over_Union()
    FWIoU = evaluator.Frequency_Weighted_Intersection_over_Union()
    log("Acc:{}, Acc_class:{}, mIoU:{}, fwIoU: {}, video mIOU: {}, video fwIOU: {}".format(
        Acc, Acc_class, mIoU, FWIoU, total_vmIOU, total_vfwIOU))
    VC_Acc = float(np.nanmean(np.array(total_VC_acc))) if len(total_VC_acc) else float("nan")
    log("Video Consistency num :{} acc:{}".format(args.vc_clip_num, VC_Acc))
    log("Inference done!")
    return {"Acc": Acc, "Acc_class": Acc_class, "mIoU": mIoU, "fwIoU": FWIoU, "video_mIoU": total_vmIOU,
            "video_fwIoU": total_vfwIOU, "VC": VC_Acc, "confusion_matrix": evaluator.confusion_matrix.copy()}


def build_parser():
    """The flags of test_clip2.py:352-401."""
    p = argparse.ArgumentParser(description="VSPW clip evaluation on the MI355X hot path")
    p.add_argument("--cfg", default="config/ade20k-hrnetv2.yaml", metavar="FILE", type=str)
    p.add_argument("--num_class", type=int, default=124)
    p.add_argument("--start_gpu", type=int, default=0)
    p.add_argument("--dataroot", type=str, default="")
    p.add_argument("--saveroot", type=str, default="")
    p.add_argument("--load_en", type=str, default="")
    p.add_argument("--load_de", type=str, default="")
    p.add_argument("--load", type=str, default="")
    p.add_argument("--batchsize", type=int, default=4)
    p.add_argument("--split", type=str, default="val")
    p.add_argument("--is_save", type=str2bool, default=False)
    p.add_argument("--lesslabel", type=str2bool, default=False)
    p.add_argument("--use_720p", type=str2bool, default=False)
    p.add_argument("--clip_num", type=int, default=5)
    p.add_argument("--dilation_num", type=int, default=0)
    p.add_argument("--gpu_num", type=int, default=1)
    p.add_argument("--propclip2", type=str2bool, default=False)
    p.add_argument("--early_usecat", type=str2bool, default=False)
    p.add_argument("--earlyfuse", type=str2bool, default=False)
    p.add_argument("--allsup", type=str2bool, default=False)
    p.add_argument("--allsup_scale", type=float, default=0.3)
    p.add_argument("--deepsup_scale", type=float, default=0.0)
    p.add_argument("--linear_combine", type=str2bool, default=False)
    p.add_argument("--distsoftmax", type=str2bool, default=False)
    p.add_argument("--distnearest", type=str2bool, default=False)
    p.add_argument("--temp", type=float, default=3)
    p.add_argument("--max_distances", type=str, default="10")
    p.add_argument("--method", type=str, default="",  # the reference lists the same methods in this order here
                   choices=["tdnet", "ETC", "nonlocal3d", "netwarp"] + [m for m in METHODS if m not in
                                                                       ("tdnet", "ETC", "nonlocal3d", "netwarp")])
    p.add_argument("--clipocr_all", type=str2bool, default=False)
    p.add_argument("--dilation2", type=str, default="2,5,9")
    p.add_argument("--use_memory", type=str2bool, default=False)
    p.add_argument("--memory_num", type=int, default=8)
    p.add_argument("--vc_clip_num", type=int, default=8)
    p.add_argument("--psp_weight", type=str2bool, default=False)
    p.add_argument("--raft_weights", type=str, default="")  # NetWarp's RAFT comes from --load; no separate file needed
    p.add_argument("opts", help="Modify config options using the command-line", default=None, nargs=argparse.REMAINDER)
    return p


if __name__ == "__main__":
    args = build_parser().parse_args()
    args.max_distances = [int(dd) for dd in args.max_distances.split(",")]
    cfg.merge_from_file(args.cfg)
    cfg.merge_from_list(args.opts)
    logger = setup_logger(distributed_rank=0)
    logger.info("Loaded configuration file {}".format(args.cfg))
    logger.info("Running with config:\n{}".format(cfg))
    cfg.MODEL.arch_encoder = cfg.MODEL.arch_encoder.lower()
    cfg.MODEL.arch_decoder = cfg.MODEL.arch_decoder.lower()
    cfg.MODEL.weights_encoder = args.load_en
    cfg.MODEL.weights_decoder = args.load_de
    main(cfg, args.start_gpu, args)
