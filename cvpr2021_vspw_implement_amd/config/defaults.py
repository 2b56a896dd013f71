"""Default configuration, mirroring the reference's config/defaults.py:1-97 without the yacs dependency.

`CfgNode` keeps the parts of yacs the drivers use: attribute access on nested nodes, `merge_from_file` (YAML),
`merge_from_list` (`KEY.SUB value` pairs from the command line), new keys may be added at run time
(train_clip2.py:519-528 adds TRAIN.max_iters / running_lr_*), `str(cfg)` dumps YAML."""
import ast
import copy

import yaml


class CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def clone(self):
        return copy.deepcopy(self)

    def freeze(self):
        pass

    def _merge(self, other, path=""):
        for k, v in other.items():
            if k not in self:
                raise KeyError("Non-existent config key: %s%s" % (path, k))
            if isinstance(self[k], CfgNode):
                if not isinstance(v, dict):
                    raise ValueError("config key %s%s is a section" % (path, k))
                self[k]._merge(v, path + k + ".")
            else:
                self[k] = _coerce(v, self[k], path + k)

    def merge_from_file(self, cfg_filename):
        with open(cfg_filename, "r") as f:
            self._merge(yaml.safe_load(f) or {})

    def merge_from_list(self, cfg_list):
        cfg_list = list(cfg_list or [])
        if len(cfg_list) % 2:
            raise ValueError("Override list has odd length: %s" % (cfg_list,))
        for full_key, v in zip(cfg_list[0::2], cfg_list[1::2]):
            node = self
            parts = full_key.split(".")
            for p in parts[:-1]:
                if p not in node:
                    raise KeyError("Non-existent config key: %s" % full_key)
                node = node[p]
            if parts[-1] not in node:
                raise KeyError("Non-existent config key: %s" % full_key)
            try:
                v = ast.literal_eval(v)
            except (ValueError, SyntaxError):
                pass
            node[parts[-1]] = _coerce(v, node[parts[-1]], full_key)

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, CfgNode) else (list(v) if isinstance(v, tuple) else v))
                for k, v in self.items()}

    def __str__(self):
        return yaml.safe_dump(self.to_dict(), default_flow_style=False)


def _coerce(value, current, key):
    """yacs-style type agreement: tuples written as strings / lists in YAML, ints promoted to floats."""
    if isinstance(current, tuple):
        if isinstance(value, str):
            value = ast.literal_eval(value)
        return tuple(value)
    if isinstance(current, float) and isinstance(value, int) and not isinstance(value, bool):
        return float(value)
    if isinstance(current, float) and isinstance(value, str):
        return float(value)
    if current is not None and not isinstance(value, type(current)) and not (
            isinstance(current, (int, float)) and isinstance(value, (int, float))):
        raise ValueError("Type mismatch for config key %s: %r vs default %r" % (key, value, current))
    return value


_C = CfgNode()
_C.DIR = "ckpt/ade20k-resnet50dilated-ppm_deepsup"

_C.DATASET = CfgNode()
_C.DATASET.root_dataset = "./data/"
_C.DATASET.list_train = "./data/training.odgt"
_C.DATASET.list_val = "./data/validation.odgt"
_C.DATASET.num_class = 150
_C.DATASET.imgSizes = (300, 375, 450, 525, 600)
_C.DATASET.imgMaxSize = 1000
_C.DATASET.padding_constant = 8
_C.DATASET.segm_downsampling_rate = 8
_C.DATASET.random_flip = True

_C.MODEL = CfgNode()
_C.MODEL.arch_encoder = "resnet50dilated"
_C.MODEL.arch_decoder = "ppm_deepsup"
_C.MODEL.weights_encoder = ""
_C.MODEL.weights_decoder = ""
_C.MODEL.fc_dim = 2048

_C.TRAIN = CfgNode()
_C.TRAIN.batch_size_per_gpu = 2
_C.TRAIN.num_epoch = 20
_C.TRAIN.start_epoch = 0
_C.TRAIN.epoch_iters = 5000
_C.TRAIN.optim = "SGD"
_C.TRAIN.lr_encoder = 0.02
_C.TRAIN.lr_decoder = 0.02
_C.TRAIN.lr_pow = 0.9
_C.TRAIN.beta1 = 0.9
_C.TRAIN.weight_decay = 1e-4
_C.TRAIN.deep_sup_scale = 0.4
_C.TRAIN.fix_bn = False
_C.TRAIN.workers = 16
_C.TRAIN.disp_iter = 20
_C.TRAIN.seed = 304

_C.VAL = CfgNode()
_C.VAL.batch_size = 1
_C.VAL.visualize = False
_C.VAL.checkpoint = "epoch_20.pth"

_C.TEST = CfgNode()
_C.TEST.batch_size = 1
_C.TEST.checkpoint = "epoch_20.pth"
_C.TEST.result = "./"
