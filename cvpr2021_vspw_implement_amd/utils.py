"""Evaluation helpers with the reference's definitions (utils.py:37-107): confusion-matrix Evaluator (pixel accuracy,
class accuracy, mIoU over classes present in the ground truth, frequency-weighted IoU) and the video-consistency
score `get_common`.  Host-side numpy, as in the reference."""
import numpy as np


class AverageMeter(object):
    def __init__(self):
        self.initialized = False
        self.val = self.avg = self.sum = self.count = None

    def update(self, val, weight=1):
        if not self.initialized:
            self.val, self.avg, self.sum, self.count, self.initialized = val, val, val * weight, weight, True
        else:
            self.val = val
            self.sum += val * weight
            self.count += weight
            self.avg = self.sum / self.count

    def value(self):
        return self.val

    def average(self):
        return self.avg


class Evaluator(object):
    def __init__(self, num_class):
        self.num_class = num_class
        self.confusion_matrix = np.zeros((num_class, num_class))

    def reset(self):
        self.confusion_matrix = np.zeros((self.num_class, self.num_class))

    def add_batch(self, gt_image, pre_image):
        assert gt_image.shape == pre_image.shape
        keep = (gt_image >= 0) & (gt_image < self.num_class)
        idx = self.num_class * gt_image[keep].astype("int") + pre_image[keep]
        self.confusion_matrix += np.bincount(idx, minlength=self.num_class ** 2).reshape(self.num_class,
                                                                                         self.num_class)

    def beforeval(self):
        present = self.confusion_matrix.sum(axis=1) > 0
        self.confusion_matrix = self.confusion_matrix * present

    def Pixel_Accuracy(self):
        return np.diag(self.confusion_matrix).sum() / self.confusion_matrix.sum()

    def Pixel_Accuracy_Class(self):
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.nanmean(np.diag(self.confusion_matrix) / self.confusion_matrix.sum(axis=1))

    def _iou(self):
        cm = self.confusion_matrix
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.diag(cm) / (cm.sum(axis=1) + cm.sum(axis=0) - np.diag(cm))

    def Mean_Intersection_over_Union(self):
        present = self.confusion_matrix.sum(axis=1) > 0
        return np.nansum(self._iou() * present) / present.sum()

    def Frequency_Weighted_Intersection_over_Union(self):
        freq = self.confusion_matrix.sum(axis=1) / self.confusion_matrix.sum()
        iu = self._iou()
        return (freq[freq > 0] * iu[freq > 0]).sum()


def get_common(gt_list, pred_list, clip_num, h, w):
    """Video consistency VC_n (reference utils.py:37-53): over every window of clip_num frames, the fraction of
    pixels whose ground truth is constant across the window that are also predicted constantly AND..."""
    accs = []
    for i in range(len(gt_list) - clip_num):
        gt_same = np.ones((h, w), dtype=bool)
        pr_same = np.ones((h, w), dtype=bool)
        for j in range(1, clip_num):
            gt_same &= gt_list[i] == gt_list[i + j]
            pr_same &= pred_list[i] == pred_list[i + j]
        accs.append((pr_same & gt_same).sum() / gt_same.sum())
    return accs


# ------------------------------------------------------------------------------------------- per-image metrics
def unique(ar, return_index=False, return_inverse=False, return_counts=False):
    """utils.py:170-210: sorted distinct values of the flattened array (numpy.unique's contract)."""
    return np.unique(np.asanyarray(ar).ravel(), return_index=return_index, return_inverse=return_inverse,
                     return_counts=return_counts)


def colorEncode(labelmap, colors, mode="RGB"):
    """utils.py:213-227: label map [h,w] -> uint8 colour image [h,w,3] through the table `colors` [n,3]; negative
    labels stay black; mode 'BGR' reverses the channel order."""
    lab = np.asarray(labelmap).astype("int")
    table = np.asarray(colors).astype(np.uint8)
    out = np.zeros(lab.shape + (3,), dtype=np.uint8)
    ok = lab >= 0
    out[ok] = table[lab[ok]]
    return out[:, :, ::-1] if mode == "BGR" else out


def accuracy(preds, label):
    """utils.py:230-235: (pixel accuracy over label >= 0, number of such pixels)."""
    valid = label >= 0
    hit = (valid * (preds == label)).sum()
    n = valid.sum()
    return float(hit) / (n + 1e-10), n


def intersectionAndUnion(imPred, imLab, numClass):
    """utils.py:238-258: per-class intersection and union areas; label -1 (unlabeled) pixels count for nothing."""
    pred = np.asarray(imPred).copy() + 1
    lab = np.asarray(imLab).copy() + 1
    pred = pred * (lab > 0)
    inter = pred * (pred == lab)
    edges = dict(bins=numClass, range=(1, numClass))
    area_i = np.histogram(inter, **edges)[0]
    area_p = np.histogram(pred, **edges)[0]
    area_l = np.histogram(lab, **edges)[0]
    return area_i, area_p + area_l - area_i


def find_recursive(root_dir, ext=".jpg"):
    """utils.py:125-132: every file under root_dir whose name ends in ext, hidden files skipped."""
    import os

    found = []
    for root, _dirs, names in os.walk(root_dir):
        found.extend(os.path.join(root, n) for n in names if n.endswith(ext) and not n.startswith("."))
    return found


# ------------------------------------------------------------------------------------------- driver conveniences
class NotSupportedCliException(ValueError):
    """utils.py:261 (a ValueError here too: callers that caught that keep working)."""


def parse_devices(input_devices):
    """utils.py:282-302: '0-3' / '0,1' / 'gpu0-gpu2' -> ['gpu0', 'gpu1', ...] without duplicates."""
    import re

    ret = []
    for d in input_devices.split(","):
        d = d.lower().strip()
        m = re.match(r"^(?:gpu)?(\d+)$", d)
        if m:
            found = [int(m.group(1))]
        else:
            m = re.match(r"^(?:gpu)?(\d+)-(?:gpu)?(\d+)$", d)
            if not m:
                raise NotSupportedCliException('Can not recognize device: "{}"'.format(d))
            a, b = int(m.group(1)), int(m.group(2))
            if a > b:
                a, b = b, a
            found = list(range(a, b + 1))
        for x in found:
            if "gpu%d" % x not in ret:
                ret.append("gpu%d" % x)
    return ret


def setup_logger(distributed_rank=0, filename="log.txt"):
    """utils.py:110-122: stdout logger on the master process only."""
    import logging
    import sys

    logger = logging.getLogger("Logger")
    logger.setLevel(logging.DEBUG)
    if distributed_rank > 0 or logger.handlers:
        return logger
    ch = logging.StreamHandler(stream=sys.stdout)
    ch.setLevel(logging.DEBUG)
    ch.setFormatter(logging.Formatter("[%(asctime)s %(levelname)s %(filename)s line %(lineno)d %(process)d] %(message)s"))
    logger.addHandler(ch)
    return logger


def vspw_palette():
    """The 256-entry palette test_clip2.py:25 writes into saved predictions: the 22 PASCAL-style colours the reference
    lists (with 191 where VOC has 192), then grey (i, i, i)."""
    pal = []
    for i in range(256):
        if i >= 22:
            pal += [i, i, i]
            continue
        c, r, g, b = i, 0, 0, 0
        for j in range(8):
            r |= ((c >> 0) & 1) << (7 - j)
            g |= ((c >> 1) & 1) << (7 - j)
            b |= ((c >> 2) & 1) << (7 - j)
            c >>= 3
        pal += [191 if v == 192 else v for v in (r, g, b)]
    return pal
