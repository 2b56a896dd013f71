"""`from models.td4_psp.loss import OhemCELoss2D` (train_clip2.py:19, :267): the hard-example loss of the TDNet method,
out of scope with it; constructing one raises."""


class OhemCELoss2D(object):
    def __init__(self, *a, **k):
        raise NotImplementedError("OhemCELoss2D belongs to --method td4_psp, which is outside the MI355X hot-path scope "
                                  "(SURVEY.md §8)")


SegmentationLosses = OhemCELoss2D
