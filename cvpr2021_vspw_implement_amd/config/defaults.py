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


# Keys, nesting and default values of the reference's config (config/defaults.py:9-97), as one literal.
_DEFAULTS = {
    "DIR": "ckpt/ade20k-resnet50dilated-ppm_deepsup",
    "DATASET": {
        "root_dataset": "./data/", "list_train": "./data/training.odgt", "list_val": "./data/validation.odgt",
        "num_class": 150,
        "imgSizes": (300, 375, 450, 525, 600),  # multiscale train/test, size of the short edge
        "imgMaxSize": 1000, "padding_constant": 8, "segm_downsampling_rate": 8, "random_flip": True,
    },
    "MODEL": {"arch_encoder": "resnet50dilated", "arch_decoder": "ppm_deepsup", "weights_encoder": "",
              "weights_decoder": "", "fc_dim": 2048},
    "TRAIN": {
        "batch_size_per_gpu": 2, "num_epoch": 20, "start_epoch": 0, "epoch_iters": 5000, "optim": "SGD",
        "lr_encoder": 0.02, "lr_decoder": 0.02, "lr_pow": 0.9, "beta1": 0.9, "weight_decay": 1e-4,
        "deep_sup_scale": 0.4, "fix_bn": False, "workers": 16, "disp_iter": 20, "seed": 304,
    },
    "VAL": {"batch_size": 1, "visualize": False, "checkpoint": "epoch_20.pth"},
    "TEST": {"batch_size": 1, "checkpoint": "epoch_20.pth", "result": "./"},
}
_C = CfgNode(_DEFAULTS)
