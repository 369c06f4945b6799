"""Time pn_mlp_forward (10 GEMMs + heads) at M = 524288 in both GEMM modes, with split-kernel ablations (GPU box)."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from pano_nerf_amd import _lib as lib
import pano_nerf_amd as pn
dev = torch.device("cuda:0"); st = torch.cuda.current_stream().cuda_stream
M = 524288; N = 128; B = M // N
model = pn.PanoMipNeRF(num_samples=N, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=5).to(dev)
if os.environ.get("PN_ZERO_WEIGHTS"):  # power / clock experiment: all-zero operands through the same instruction stream
    with torch.no_grad():
        model.mlp.flat_params().zero_()
    from pano_nerf_amd.mlp import mark_dirty
    mark_dirty(model.mlp)
mean = torch.randn(M, 3, device=dev); cov = torch.rand(M, 3, device=dev) * 1e-3; vd = torch.nn.functional.normalize(torch.randn(B, 3, device=dev), dim=-1)
Mp = int(lib.load().pn_pad_rows(M))
E = lambda *s: torch.empty(*s, device=dev)
enc, venc, vb, acts, rr, rd = E(Mp, 96), E(B, 27), E(B, 128), E(10, Mp, 256), E(M, 3), E(M, 5)
masks = torch.empty(9, Mp, 8, dtype=torch.int32, device=dev)
def run():
    lib.call("pn_mlp_forward", M, N, B, 5, flat.data_ptr(), wpack.data_ptr(), mean.data_ptr(), cov.data_ptr(), vd.data_ptr(), enc.data_ptr(),
             venc.data_ptr(), vb.data_ptr(), acts.data_ptr(), masks.data_ptr(), rr.data_ptr(), rd.data_ptr(), st)
def timeit(n=5):
    for _ in range(2): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
flop = M * 1222656.0
for mode, dbg, name in ((0, 0, "fp32 mfma"), (1, 0, "split (all-DMA ring kernel)"), (1, 0x8000, "ring, no MFMAs"), (1, 0x10000, "ring, no split arithmetic"),
                        (1, 0x18000, "ring, neither"), (1, 0x1800, "ring, no DMA"),
                        (1, 256, "split, narrow kernel only")):
    lib.load().pn_set_gemm_mode(mode)
    lib.load().pn_prof_enable(dbg << 8)
    flat = model.mlp.flat_params(); wpack = model.mlp.packed(st)
    ms = timeit()
    print(f"{name:30s} {ms:7.3f} ms  {flop/ms/1e9:7.1f} TF-equivalent")
