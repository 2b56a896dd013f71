"""The four SGD parameter-group generators shared by every clip head.

Reference semantics (models/clip_psp.py:99-135, models/clip_ocr.py:72-102, models/non_local_models.py:81-112,
models/netwarp.py:116-149): walk `named_modules()` of each listed module and, for EVERY sub-module, its recursive
`named_parameters()`; yield parameters whose name does / does not contain 'bias'.  A parameter is therefore yielded
once per ancestor module inside the listed root (the reference relies on torch.optim tolerating duplicates); the
multiplicity is part of the training recipe and is preserved here.
"""


def iter_params(roots, bias):
    for root in roots:
        for _, sub in root.named_modules():
            for key, p in sub.named_parameters():
                if p.requires_grad and (("bias" in key) == bias):
                    yield p


class LrGroupsMixin:
    """Expects `_lr_1x_roots()` / `_lr_10x_roots()` / `_lr_10x_bias_roots()` on the host class."""

    def get_1x_lr_params(self):
        return iter_params(self._lr_1x_roots(), bias=False)

    def get_10x_lr_params(self):
        return iter_params(self._lr_10x_roots(), bias=False)

    def get_1x_lr_params_bias(self):
        return iter_params(self._lr_1x_roots(), bias=True)

    def get_10x_lr_params_bias(self):
        return iter_params(self._lr_10x_bias_roots(), bias=True)

    def _lr_1x_roots(self):
        return [self.encoder]

    def _lr_10x_bias_roots(self):
        return self._lr_10x_roots()
