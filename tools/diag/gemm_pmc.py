"""One short-K GEMM shape in a loop, for rocprofv3 --pmc passes (tools/diag/gemm_pmc.sh)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from cvpr2021_vspw_implement_amd import _C
dev = torch.device("cuda:0")
B, M, N, K = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (16, 9000, 256, 256))]
a = torch.randn(B, M, K, device=dev); b = torch.randn(B, N, K, device=dev); c = torch.empty(B, M, N, device=dev)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(6):
    _C.call("vspw_bmm_nt", a.data_ptr(), b.data_ptr(), c.data_ptr(), B, M, N, K, st)
torch.cuda.synchronize()
