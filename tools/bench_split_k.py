"""Split-mode NT GEMM (256-wide layer weights, M = 524288) as a function of K: time = fixed part (launch ramp, epilogue)
+ K x main-loop cost.  Uses layer-1 weights of a real model so that the pre-split bf16 planes exist."""
import os, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from pano_nerf_amd import _lib as lib
import pano_nerf_amd as pn
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
M = 524288
model = pn.PanoMipNeRF(num_samples=128, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=5).to(dev)
off = (ctypes.c_int64 * 24)()
lib.load().pn_param_layout(5, off)
A = torch.relu(torch.randn(M, 256, device=dev)); C = torch.empty(M, 256, device=dev); bias = torch.randn(256, device=dev)
def timeit(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for mode, dbg, name in ((0, 0, "fp32 mfma"), (1, 0, "split (narrow kernel: a free-standing B has no pre-split planes)")):
    lib.load().pn_set_gemm_mode(mode)
    lib.load().pn_prof_enable(dbg << 8)
    flat = model.mlp.flat_params(); wpack = model.mlp.packed(st)
    W1 = flat.data_ptr() + 4 * off[2]  # layer-1 weight [256][256]
    row = []
    for K in (16, 64, 128, 256):
        for flags, tag in ((3, ""), (3 | 0x100, "ns")):
            us = timeit(lambda: lib.call("pn_gemm_nt", M, 256, K, A.data_ptr(), 256, W1, 256, C.data_ptr(), 256, bias.data_ptr(), None, 256, flags, st))
            row.append(f"K={K}{tag}: {us:6.1f}us")
    print(f"{name:18s} " + "  ".join(row), flush=True)
