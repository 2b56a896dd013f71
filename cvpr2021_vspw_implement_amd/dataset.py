"""`from dataset import TrainDataset` (train_clip2.py:15): the ADE20K-style .odgt dataset of the code base the
reference grew out of - imported by its drivers, never constructed by the VSPW scripts (`scripts/*.sh` all use the
dataset2 classes).  Importable, raises at construction."""


class TrainDataset(object):
    def __init__(self, *a, **k):
        raise NotImplementedError("dataset.TrainDataset (.odgt lists) is not part of the VSPW path; use dataset2.*")


ValDataset = TestDataset = TrainDataset
