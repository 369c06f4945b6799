"""Accuracy + speed of the 3-term bf16-split NT GEMM vs the fp32-MFMA one (GPU box)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from pano_nerf_amd import _lib as lib
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
gen = torch.Generator().manual_seed(0)
for (M, N, K) in ((300, 256, 96), (1000, 128, 256), (129, 96, 352), (4096, 256, 256)):
    A = torch.randn(M, K, generator=gen); Bt = torch.randn(N, K, generator=gen) * 0.06
    A[0, :4] = torch.tensor([1e-8, 3e4, -7.5e-3, 1.0])
    ref = (A.double() @ Bt.double().T)
    dA, dB = A.to(dev), Bt.to(dev)
    outs = {}
    for mode in (0, 1):
        lib.load().pn_set_gemm_mode(mode)
        C = torch.empty(M, N, device=dev)
        lib.call("pn_gemm_nt", M, N, K, dA.data_ptr(), K, dB.data_ptr(), K, C.data_ptr(), N, None, None, 0, 0, st)
        torch.cuda.synchronize()
        outs[mode] = C.cpu().double()
    sc = ref.abs().max()
    print(f"M{M} N{N} K{K}: fp32 max err/scale {float((outs[0]-ref).abs().max()/sc):.2e}  split {float((outs[1]-ref).abs().max()/sc):.2e}  "
          f"rms fp32 {float((outs[0]-ref).pow(2).mean().sqrt()/sc):.2e} split {float((outs[1]-ref).pow(2).mean().sqrt()/sc):.2e}")
M = 524288
A = torch.randn(M, 256, device=dev); W = torch.randn(256, 256, device=dev) * 0.06; C = torch.empty(M, 256, device=dev); bias = torch.randn(256, device=dev)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for mode in (0, 1):
    lib.load().pn_set_gemm_mode(mode)
    for name, N, K, flags in (("256x256 plain", 256, 256, 0), ("256x256 bias+relu", 256, 256, 3), ("K=96", 256, 96, 3), ("N=96", 96, 256, 0)):
        ms = timeit(lambda: lib.call("pn_gemm_nt", M, N, K, A.data_ptr(), 256, W.data_ptr(), 256, C.data_ptr(), 256, bias.data_ptr(), None, 256, flags, st))
        print(f"mode {mode} nt {name:18s} {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TF-equivalent")
