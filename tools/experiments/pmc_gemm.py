"""Run a few launches of each GEMM kernel (for rocprofv3 --pmc)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R)
import torch
from pano_nerf_amd import _lib as lib
dev = torch.device("cuda:0"); M = 524288
st = torch.cuda.current_stream().cuda_stream
A = torch.randn(M, 256, device=dev); W = torch.randn(256, 256, device=dev) * 0.06
C = torch.empty(M, 256, device=dev); bias = torch.randn(256, device=dev)
for flags in (0, 0x300):
    for _ in range(3):
        lib.call("pn_gemm_nt", M, 256, 256, A.data_ptr(), 256, W.data_ptr(), 256, C.data_ptr(), 256, bias.data_ptr(), None, 256, flags, st)
X = torch.randn(M, 256, device=dev); Cw = torch.zeros(256, 256, device=dev)
work = torch.empty(int(lib.load().pn_gemm_tn_work_floats(M, 256, 256)), device=dev)
for _ in range(3):
    lib.call("pn_gemm_tn", M, 256, 256, X.data_ptr(), 256, A.data_ptr(), 256, Cw.data_ptr(), 256, 1, work.data_ptr(), st)
torch.cuda.synchronize()
