"""The `pixel_acc` method every segmentation module of the reference carries (models/models.py:65-71 and the copies in
clip_psp.py, clip_ocr.py, netwarp.py, non_local_models.py): kept as public surface - the training paths here get the same
number out of the fused loss kernel (csrc/loss.hip: seg_nll) and never call it."""


def pixel_accuracy(pred, label):
    """Share of the pixels with label >= 0 whose highest-scoring class is the label.  pred [N, K, H, W] scores, label
    [N, H, W] integer classes; the ignore value 255 is >= 0 and therefore counts as a (never matched) valid pixel - the
    reference's behaviour, which the fused kernel reproduces."""
    valid = label.ge(0)
    hits = pred.argmax(dim=1).eq(label).logical_and(valid)
    return hits.sum().float() / (valid.sum().float() + 1e-10)
