"""Order dependence of the TCB-OCR trajectory test (it passed alone and failed after the TCB-PSP cases): run the cases in
one process in a given order and print |loss - fp64| of the first steps.  usage: traj_order.py psp:0 psp:1 ocr:0 [drop]"""
import os, sys, tempfile, pathlib, gc
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_drivers_gpu as TD
from helpers import golden
from cvpr2021_vspw_implement_amd import ops
dev = torch.device("cuda:0")
for spec in sys.argv[1:]:
    if spec == "drop":
        ops.drop_weight_transpose_cache(); print("dropped derived-weight caches"); continue
    if spec.startswith("clear:"):
        from cvpr2021_vspw_implement_amd import _ops_conv as OC
        which = spec.split(":")[1]
        {"wt": OC._wt_copies, "wu": OC._wu_copies, "wu3": OC._wu3_copies[3], "wu4": OC._wu3_copies[4]}[which].clear()
        print("cleared", which); continue
    if spec == "fold":
        ops.invalidate_inference_cache(); print("invalidated inference cache"); continue
    if spec == "gc":
        gc.collect(); torch.cuda.empty_cache(); print("gc"); continue
    kind, g = spec.split(":")
    kind = {"psp": "clip_psp", "ocr": "clip_ocr"}[kind]
    tmp = pathlib.Path(tempfile.mkdtemp())
    try:
        TD.test_tcb_training_trajectory_follows_the_reference(dev, tmp, kind, g == "1")
        print(spec, "PASSED")
    except AssertionError as e:
        print(spec, "FAILED", str(e)[:200])
