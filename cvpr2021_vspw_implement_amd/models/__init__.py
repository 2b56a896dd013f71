"""Drop-in `models` package: exports every name the reference drivers import (train_clip2.py:14-21,
test_clip2.py:13-22).  Heads outside the MI355X hot-path scope are stubs that raise at construction."""
from .models import (ModelBuilder, SegmentationModule, SegmentationModuleBase, ClipWarpNet, Resnet, ResnetDilated,  # noqa: F401
                     PPM, PPMDeepsup, PPMDeepsup_clip, PPM_clip, _stub)
from .clip_psp import Clip_PSP, PPM_conv  # noqa: F401
from .clip_ocr import ClipOCRNet  # noqa: F401
from .ocrnet import SpatialOCRNet  # noqa: F401
from .non_local import NLBlockND  # noqa: F401
from .non_local_models import Non_local2d, Non_local3d  # noqa: F401
from .netwarp import NetWarp, NetWarp_ocr, FlowCNN, flowwarp  # noqa: F401

ETC = _stub("ETC")
ETC_ocr = _stub("ETC_ocr")
PropNet = _stub("PropNet")
OurWarpMerge = _stub("OurWarpMerge")
WarpNet = _stub("WarpNet")
