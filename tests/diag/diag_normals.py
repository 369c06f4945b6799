"""Diagnostic (GPU box): error of pn_density_grad vs the fp32 / fp64 oracle, per sample."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from conftest import load_golden
from oracle import pano_oracle as orc
from pano_nerf_amd import _lib as lib
import test_gpu_stages as T

for case in ("B64_N32", "B16_N128"):
    g = load_golden("stages_" + case)
    B, N = g["mean_rnd"].shape[:2]; M = B * N
    params = orc.init_params(4, 5)
    flat, wpack, _, _ = T._flat_params(lib, params, 5)
    mean, cov, vd = T.G(g["mean_rnd"]).view(M, 3), T.G(g["cov_rnd"]).view(M, 3), T.G(g["ray_viewdirs"])
    buf = T._mlp_eval(lib, flat, wpack, 5, mean, cov, vd, N)
    Mp = int(lib.load().pn_pad_rows(M))
    rs, scratch, gm = T.E(8, Mp, 256), T.E(Mp, 96), T.E(M, 3)
    lib.call("pn_density_grad", M, 5, -1.0, flat.data_ptr(), wpack.data_ptr(), mean.data_ptr(), cov.data_ptr(),
             buf["acts"].data_ptr(), buf["masks"].data_ptr(), buf["raw_den"].data_ptr(), rs.data_ptr(), scratch.data_ptr(), gm.data_ptr(), T.st())
    got = T.C(gm).view(B, N, 3).double()
    def oracle(dt):
        torch.set_default_dtype(dt)
        try:
            p = {k: v.to(dt) for k, v in params.items()}
            m = torch.from_numpy(g["mean_rnd"]).to(dt).requires_grad_(True)
            _, sig, _ = orc.radiance_field(p, m, torch.from_numpy(g["cov_rnd"]).to(dt), torch.from_numpy(g["ray_viewdirs"]).to(dt))
            (r,) = torch.autograd.grad(sig.sum(), m)
            acts = None
        finally:
            torch.set_default_dtype(torch.float32)
        return r.double()
    r32, r64 = oracle(torch.float32), oracle(torch.float64)
    sc = r64.abs().max()
    for name, a in (("ours", got), ("oracle32", r32)):
        e = (a - r64).abs()
        print(case, name, "max/scale %.3e" % float(e.max() / sc), "median rel %.3e" % float((e / (r64.abs() + 1e-9)).median()),
              "n(e>1e-4*sc)=%d" % int((e > 1e-4 * sc).sum()))
    e = (got - r64).abs().amax(-1).view(-1)
    top = torch.topk(e, 5).indices
    for i in top:
        b, n = divmod(int(i), N)
        print("  sample", b, n, "ours", got[b, n].numpy(), "o32", r32[b, n].numpy(), "o64", r64[b, n].numpy(), "cov", g["cov_rnd"][b, n])
    # encoding-gradient check: compare d sigma/d enc
    print("  scale", float(sc))
