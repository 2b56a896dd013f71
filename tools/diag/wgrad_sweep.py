"""DIAGNOSTIC: weight-gradient time (GEMM + split-K reduce) against the number of pixel chunks, per shape.
Run once per candidate: VSPW_WGRAD_SPLITS=<s> python tools/diag/wgrad_sweep.py (0 / unset = the library's own plan)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cvpr2021_vspw_implement_amd import ops
dev = torch.device("cuda:0")
ops.set_wgrad_side_stream(False)
SHAPES = [("3x3 256", 10, 60, 256, 256, 3, 2, 2), ("1x1 256->1024", 10, 60, 256, 1024, 1, 0, 1),
          ("1x1 1024->256", 10, 60, 1024, 256, 1, 0, 1), ("3x3 512 d4", 10, 60, 512, 512, 3, 4, 4),
          ("deepsup 1024->512", 10, 60, 1024, 512, 3, 1, 1), ("conv_last 4096->512", 2, 60, 4096, 512, 3, 1, 1),
          ("1x1 512->2048", 10, 60, 512, 2048, 1, 0, 1), ("1x1 2048->512", 10, 60, 2048, 512, 1, 0, 1),
          ("stem 64->128 240", 10, 240, 64, 128, 3, 1, 1), ("stem 64->64 240", 10, 240, 64, 64, 3, 1, 1),
          ("l2 3x3 128", 10, 60, 128, 128, 3, 1, 1), ("l1 3x3 64 120", 10, 120, 64, 64, 3, 1, 1),
          ("l2 1x1 128->512", 10, 60, 128, 512, 1, 0, 1), ("l2 1x1 512->128", 10, 60, 512, 128, 1, 0, 1)]
flt = sys.argv[1] if len(sys.argv) > 1 else ""
out = []
for name, n, hw, c, k, ks, pad, dil in SHAPES:
    if flt and flt not in name:
        continue
    x = ops.empty_nhwc(n, c, hw, hw, dev).normal_()
    wt = (torch.randn(k, ks, ks, c, device=dev) * 0.05).permute(0, 3, 1, 2)
    y, part, d = ops.conv2d_forward(x, wt, None, 1, pad, dil, want_stats=False)
    dy = torch.randn_like(y)
    for _ in range(3):
        ops.conv2d_backward_weight(dy, x, d)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv2d_backward_weight(dy, x, d)
    e1.record()
    torch.cuda.synchronize()
    out.append("%s %.4f" % (name.replace(" ", "_"), e0.elapsed_time(e1) / 20))
print("splits=%s " % os.environ.get("VSPW_WGRAD_SPLITS", "plan") + " ".join(out))
