"""Diagnostic (GPU box): where does the level-1 normal deviate?  Compares our intermediates with the
oracle run in fp32 and fp64 on the same inputs."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from conftest import load_golden
from oracle import pano_oracle as orc
import test_gpu_full as F

def run_oracle(dt, s, N, env):
    torch.set_default_dtype(dt)
    try:
        p = {k: v.to(dt) for k, v in orc.init_params(4, 5).items()}
        rays = orc.Rays(*[torch.from_numpy(s["ray_" + k]).to(dt) for k in orc.Rays._fields])
        t0, (m0, c0) = orc.sample_along_rays(rays.origins, rays.directions, rays.radii, N, rays.near, rays.far)
        rgb, sig, _ = orc.radiance_field(p, m0, c0, rays.viewdirs)
        _, _, _, w0 = orc.volumetric_rendering(rgb, sig, t0, rays.directions, False)
        t1, (m1, c1) = orc.resample_along_rays(rays.origins, rays.directions, rays.radii, t0, w0.detach().clone(), 0.01)
        rgb, sig, alb = orc.radiance_field(p, m1, c1, rays.viewdirs)
        comp, dist, acc, w1 = orc.volumetric_rendering(rgb, sig, t1, rays.directions, False)
        g = -orc.density_normals(p, m1, c1, rays.viewdirs)
        nrm = torch.nn.functional.normalize(-g, dim=-1)
        nw = w1[..., None] / w1.sum(-1).view(-1, 1, 1)
        normal = torch.nn.functional.normalize((nw * nrm).sum(1), dim=-1)
    finally:
        torch.set_default_dtype(torch.float32)
    return dict(t0=t0, w0=w0, t1=t1, m1=m1, c1=c1, w1=w1, g=g, normal=normal, sig=sig)

for case in ("B64_N32",):
    s = load_golden("stages_" + case); gfull = load_golden("pano_full_" + case)
    N = s["t_det"].shape[1] - 1
    rays, env = F.to_dev(F.rays_of(s)), F.to_dev(orc.Rays(*[torch.from_numpy(load_golden("raygen_8x16")["env_" + k]) for k in orc.Rays._fields]))
    model = F.make_pano(N); model.mlp.debug_keep = True
    with torch.no_grad():
        outs = model(rays=rays, env_rays=env, randomized=False, white_bkgd=False, enable_surf=True, use_ort_loss=True)
    torch.cuda.synchronize()
    o, d, vd, env_d, env_omega, e0, e1, ee, w1, env_rgb, albedo, normal, params, wpack = model.mlp.debug_pack
    B = o.shape[0]
    r32, r64 = run_oracle(torch.float32, s, N, env), run_oracle(torch.float64, s, N, env)
    ours = dict(t0=e0.t.cpu(), t1=e1.t.cpu(), m1=e1.mean.cpu().view(B, N, 3), c1=e1.cov.cpu().view(B, N, 3), w1=w1.cpu(),
                g=e1.gmean.cpu().view(B, N, 3), normal=normal.cpu())
    for k in ours:
        a, b32, b64 = ours[k].double(), r32[k].detach().double(), r64[k].detach().double()
        sc = b64.abs().max()
        print(f"{k:7s} ours-vs-64 max {float((a-b64).abs().max()/sc):.3e}   o32-vs-64 max {float((b32-b64).abs().max()/sc):.3e}   ours-vs-o32 {float((a-b32).abs().max()/sc):.3e}")
    e = (ours["g"].double() - r64["g"].double()).abs().amax(-1)
    idx = torch.topk(e.view(-1), 8).indices
    for i in idx:
        b, n = divmod(int(i), N)
        print("  g sample", b, n, "err", float(e[b, n]), "ours", ours["g"][b, n].numpy(), "o32", r32["g"][b, n].detach().numpy(), "o64", r64["g"][b, n].detach().numpy(), "w", float(r64["w1"][b, n]), "cov", ours["c1"][b,n].numpy())
    en = (ours["normal"].double() - r64["normal"].double()).abs().amax(-1)
    print("normal worst rays", torch.topk(en, 5))
