"""Time the IMPORTED reference's training step on this container's CPU and the
oracle's 'faithful' mode on the same inputs (calibration ratio).  Build container only; writes
tests/golden/ref_cpu_timing.json.  Same import shim as make_golden.py."""
import json, os, sys, time, types
sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference"); sys.path.insert(1, ROOT)
for name in ("cv2", "Imath"):
    sys.modules[name] = types.ModuleType(name)
_exr = types.ModuleType("OpenEXR"); _exr.InputFile = _exr.OutputFile = _exr.Header = object; sys.modules["OpenEXR"] = _exr
import numpy as np, torch
import models.pano_mip_nerf as rpano, models.mip_nerf as rmip
from datasets.base_datasets import Rays
sys.path.insert(0, HERE)
from make_golden import ref_loss_pano, ref_loss_mip, load_params
from oracle import pano_oracle as orc

def med(f, n=3):
    f(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))

res = {"nproc": os.cpu_count(), "torch_threads": torch.get_num_threads(), "torch": torch.__version__, "runs": []}
flat, rgbs, radius, _ = orc.synthetic_scene(16, 32, 3, seed=4)
env16 = orc.generate_lit_rays(10, radius)
envf = Rays(*[x.float() for x in env16])
for model, B, N in (("pano", 64, 128), ("pano", 128, 64), ("mip", 256, 32), ("mip", 128, 128)):
    idx = torch.arange(0, B * 3, 3)
    rays_o = orc.Rays(*[x[idx] for x in flat]); rays_r = Rays(*rays_o); gt = rgbs[idx]
    if model == "pano":
        params = orc.init_params(4, 5)
        net = rpano.PanoMipNeRF(num_samples=N, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=5, num_env_samples=10)
        load_params(net.mlp, params)
        def ref_step():
            net.zero_grad()
            outs = net(rays=rays_r, env_rays=envf, randomized=True, white_bkgd=False, enable_surf=True, use_ort_loss=True)
            ref_loss_pano(outs, rays_r.lossmult, gt).backward()
        p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        def orc_step(mode):
            gen = torch.Generator().manual_seed(0)
            noise = dict(t_rand=torch.rand(B, N + 1, generator=gen), u_rand=torch.rand(B, N + 1, generator=gen) * (1 / (N + 1) - 1.2e-7), env_rand=torch.rand(1, 11, generator=gen))
            outs = orc.pano_forward(p, rays_o, env16, num_samples=N, noise=noise, normals_mode=mode)
            torch.autograd.grad(orc.pano_loss(outs, rays_o.lossmult, gt), list(p.values()))
    else:
        params = orc.init_params(4, 1)
        net = rmip.MipNeRF(num_samples=N, rgb_activation="softplus", rgb_padding=0, mlp_num_density_channels=1)
        load_params(net.mlp, params)
        def ref_step():
            net.zero_grad()
            outs = net(rays=rays_r, randomized=True, white_bkgd=False, use_ort_loss=False)
            ref_loss_mip(outs, rays_r.lossmult, gt, False).backward()
        p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        def orc_step(mode):
            gen = torch.Generator().manual_seed(0)
            noise = dict(t_rand=torch.rand(B, N + 1, generator=gen), u_rand=torch.rand(B, N + 1, generator=gen) * (1 / (N + 1) - 1.2e-7))
            outs = orc.mip_forward(p, rays_o, num_samples=N, noise=noise)
            torch.autograd.grad(orc.mip_loss(outs, rays_o.lossmult, gt), list(p.values()))
    tr = med(ref_step); tf = med(lambda: orc_step("faithful")); tq = med(lambda: orc_step("fast"))
    res["runs"].append(dict(model=model, B=B, N=N, reference_rays_per_s=B / tr, oracle_faithful_rays_per_s=B / tf,
                            oracle_fast_rays_per_s=B / tq, oracle_faithful_over_reference=tr / tf))
    print(res["runs"][-1], flush=True)
json.dump(res, open(os.path.join(HERE, "ref_cpu_timing.json"), "w"), indent=1)
