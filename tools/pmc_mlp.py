"""A few pn_mlp_forward calls in one GEMM mode (argv[1]: 0 fp32 / 1 split) for rocprofv3 --pmc passes."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from pano_nerf_amd import _lib as lib
import pano_nerf_amd as pn
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 1
M = 524288; N = 128; B = M // N
model = pn.PanoMipNeRF(num_samples=N, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=5).to(dev)
mean = torch.randn(M, 3, device=dev); cov = torch.rand(M, 3, device=dev) * 1e-3
vd = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1)
Mp = int(lib.load().pn_pad_rows(M))
E = lambda *s: torch.empty(*s, device=dev)
enc, venc, vb, acts, rr, rd = E(Mp, 96), E(B, 27), E(B, 128), E(10, Mp, 256), E(M, 3), E(M, 5)
masks = torch.empty(9, Mp, 8, dtype=torch.int32, device=dev)
lib.load().pn_set_gemm_mode(mode)
flat = model.mlp.flat_params(); wpack = model.mlp.packed(st)
for _ in range(3):
    lib.call("pn_mlp_forward", M, N, B, 5, flat.data_ptr(), wpack.data_ptr(), mean.data_ptr(), cov.data_ptr(), vd.data_ptr(),
             enc.data_ptr(), venc.data_ptr(), vb.data_ptr(), acts.data_ptr(), masks.data_ptr(), rr.data_ptr(), rd.data_ptr(), st)
torch.cuda.synchronize()
